"""Redzone allocator for the GPU tests (test infrastructure).

Every parity test hands the kernels exact-size torch tensors from the caching allocator: a kernel that writes one row past its
output lands in allocator slack (blocks are rounded up to 512 bytes, freed blocks are recycled) and the test still passes.  Inside
`guarded()` every `torch.empty(...)` on a HIP device -- which is how the host layer of articulated-pose_amd allocates EVERY output and
scratch buffer it passes through the C ABI -- is carved out of a larger allocation instead:

      [ guard: PAD bytes of 0xA5 | body: the tensor, pre-filled with 0xFF bytes | guard: PAD bytes of 0xA5 (+ alignment slack) ]

`check()` (called when the context exits) synchronises and asserts every guard byte.  The body pattern is a NaN for float32 /
float64 and -1 for int32, so an output element a kernel failed to write shows up in the test's own equality check.
`misalign`: extra bytes in front of the body (a multiple of the element size) for the entry points that promise no alignment;
entry points that require 16-byte alignment keep misalign = 0 (PAD is a multiple of 512).
"""
import contextlib

import torch

PAD = 4096
GUARD, BODY = 0xA5, 0xFF


class RedzoneError(AssertionError):
    pass


class Arena(object):
    def __init__(self, misalign=0):
        self.blocks = []            # (raw uint8 tensor, body offset, body bytes, description)
        self.misalign = int(misalign)
        self._empty = torch.empty

    def empty_like(self, t, **kw):
        return self.empty(tuple(t.shape), dtype=kw.get("dtype", t.dtype), device=kw.get("device", t.device))

    def empty(self, *size, **kw):
        dev = kw.get("device")
        dtype = kw.get("dtype") or torch.get_default_dtype()
        if dev is None or torch.device(dev).type != "cuda" or kw.get("pin_memory") or kw.get("memory_format") not in (None, torch.contiguous_format):
            return self._empty(*size, **kw)
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(int(s) for s in size)
        item = torch.empty((), dtype=dtype).element_size()
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * item
        mis = (self.misalign // item) * item
        off = PAD + mis
        raw = self._empty((off + nbytes + PAD + 64,), dtype=torch.uint8, device=dev)
        raw.fill_(GUARD)
        raw[off:off + nbytes].fill_(BODY)
        t = raw[off:off + nbytes].view(dtype).view(shape)
        self.blocks.append((raw, off, nbytes, "%s %s" % (str(dtype), shape)))
        return t

    def check(self):
        torch.cuda.synchronize()
        bad = []
        for raw, off, nbytes, what in self.blocks:
            head, tail = raw[:off], raw[off + nbytes:]
            if not bool((head == GUARD).all()):
                i = int((head != GUARD).nonzero()[-1])
                bad.append("%s: %d byte(s) BEFORE the buffer overwritten (nearest: %d bytes before its start)" % (what, int((head != GUARD).sum()), off - i))
            if not bool((tail == GUARD).all()):
                i = int((tail != GUARD).nonzero()[0])
                bad.append("%s: %d byte(s) PAST the buffer overwritten (first: %d bytes past its end)" % (what, int((tail != GUARD).sum()), i))
        n = len(self.blocks)
        self.blocks = []
        if bad:
            raise RedzoneError("redzone violated:\n  " + "\n  ".join(bad))
        return n


@contextlib.contextmanager
def guarded(misalign=0):
    """with guarded() as arena: ...   -- patches torch.empty for the duration; arena.check() runs on a clean exit (and can be called
    earlier; it returns the number of buffers it verified)."""
    arena = Arena(misalign)
    orig, orig_like = torch.empty, torch.empty_like
    torch.empty, torch.empty_like = arena.empty, arena.empty_like
    try:
        yield arena
    finally:
        torch.empty, torch.empty_like = orig, orig_like
    arena.check()
