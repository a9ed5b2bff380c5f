"""Counterparts of lib/aligning.py on the MI355X, batched: estimateSimilarityUmeyama (:580-622) and the 5-point RANSAC
estimateSimilarityTransform (:17-32 with set_config / getRANSACInliers / evaluateModel)."""
import numpy as np
import torch

from .. import _lib


def umeyama_batch(sources, targets, device="cuda:0"):
    """sources/targets: lists of (n_i,3) arrays -> list of (Scales(3), Rotation(3,3), Translation(3), OutTransform(4,4))
    with the reference's conventions (Rotation is the TRANSPOSE of the src->tgt rotation).
    Precision: the points cross the C ABI as float32 (what the prediction records hold: P, nocs_gt are float32 datasets);
    means, covariance, rotation and scale are then computed in float64 like numpy does on float32 inputs promoted by the
    reference's float64 intermediates.  float64 inputs are therefore quantised to float32 first (~1e-7 relative)."""
    off = np.zeros(len(sources) + 1, np.int32)
    off[1:] = np.cumsum([len(s) for s in sources])
    src = torch.from_numpy(np.concatenate([np.asarray(s, np.float32).reshape(-1, 3) for s in sources])).to(device)
    tgt = torch.from_numpy(np.concatenate([np.asarray(t, np.float32).reshape(-1, 3) for t in targets])).to(device)
    offd = torch.from_numpy(off).to(device)
    out = torch.empty((len(sources), 32), dtype=torch.float64, device=device)
    _lib.call("ancsh_umeyama", len(sources), _lib.ptr(offd), _lib.ptr(src), _lib.ptr(tgt), _lib.ptr(out))
    o = out.cpu().numpy()
    return [(o[i, 0:3].copy(), o[i, 3:12].reshape(3, 3).copy(), o[i, 12:15].copy(), o[i, 15:31].reshape(4, 4).copy())
            for i in range(len(sources))]


def estimateSimilarityUmeyama(SourceHom, TargetHom, rt_pre=None):
    """Same call shape as the reference: (3|4, n) arrays -> Scales, Rotation, Translation, OutTransform."""
    if rt_pre is not None:
        raise NotImplementedError("rt_pre is never passed on the evaluation path (compute_gt_pose.py:87)")
    return umeyama_batch([np.asarray(SourceHom)[:3].T], [np.asarray(TargetHom)[:3].T])[0]


def estimate_similarity_transform_batch(sources, targets, draws=None, seed=0, niter=100, device="cuda:0"):
    """lists of (n_i,3) arrays -> list of (Scales, Rotation, Translation, OutTransform) or (None,)*4 where the
    reference returns None (BestInlierRatio < 0.1).  draws: (nprob, niter, 5) int array replaying
    np.random.randint(n, size=5) per iteration, or None for the on-device generator."""
    off = np.zeros(len(sources) + 1, np.int32)
    off[1:] = np.cumsum([len(s) for s in sources])
    src = torch.from_numpy(np.concatenate([np.asarray(s, np.float32).reshape(-1, 3) for s in sources])).to(device)
    tgt = torch.from_numpy(np.concatenate([np.asarray(t, np.float32).reshape(-1, 3) for t in targets])).to(device)
    offd = torch.from_numpy(off).to(device)
    d = None if draws is None else torch.from_numpy(np.ascontiguousarray(draws, np.int32)).to(device)
    out = torch.empty((len(sources), 32), dtype=torch.float64, device=device)
    status = torch.empty((len(sources),), dtype=torch.int32, device=device)
    _lib.call("ancsh_estimate_similarity_transform", len(sources), _lib.ptr(offd), _lib.ptr(src), _lib.ptr(tgt), int(niter),
              _lib.ptr(d), int(seed), _lib.ptr(out), _lib.ptr(status))
    o, st = out.cpu().numpy(), status.cpu().numpy()
    res = []
    for i in range(len(sources)):
        if st[i] != 0:
            res.append((None, None, None, None))
        else:
            res.append((o[i, 0:3].copy(), o[i, 3:12].reshape(3, 3).copy(), o[i, 12:15].copy(), o[i, 15:31].reshape(4, 4).copy()))
    return res


def estimateSimilarityTransform(source, target, rt_pre=None, verbose=False, draws=None, seed=0):
    """Same call shape as lib/aligning.py:17: (n,3), (n,3) -> Scales, Rotation, Translation, OutTransform (or 4 x None)."""
    if rt_pre is not None:
        raise NotImplementedError("rt_pre is never passed by the reference's entry points")
    return estimate_similarity_transform_batch([source], [target], None if draws is None else np.asarray(draws)[None], seed)[0]
