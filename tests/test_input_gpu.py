"""GPU: the batched input sampling kernel (ancsh_input_sample through articulated_pose_amd.dataset) against the golden
records produced by the reference's own Dataset.create_unit_data_from_hdf5 -- every output bit for bit (a gather, one float32
multiply, two masks) -- for a ragged batch holding all cases at once, and in device-RNG mode through its invariants."""
import os

import numpy as np
import pytest
import torch

from test_input_cpu import G, OUT_KEYS, cases, load_case

pytestmark = pytest.mark.gpu


def test_input_sample_batch_equals_reference_records(dev):
    from articulated_pose_amd.dataset import create_unit_data_batch
    with np.load(G) as z:
        loaded = {t: load_case(z, t) for t in cases()}
    for N in (1024, 2048):
        tags = [t for t in loaded if loaded[t][1] == N]
        K = max(loaded[t][3] for t in tags)                       # one launch, widest mask; narrower clouds compare their columns
        out = create_unit_data_batch([loaded[t][0] for t in tags], N, [loaded[t][2] for t in tags], K,
                                     perms=[loaded[t][4] for t in tags], device=dev)
        for b, t in enumerate(tags):
            want, k_t = loaded[t][5], loaded[t][3]
            for key in OUT_KEYS:
                got = out[key][b].cpu().numpy()
                if key == "mask_array":
                    assert np.array_equal(got[:, :k_t], want[key]) and not got[:, k_t:].any(), (t, key)
                else:
                    assert got.dtype == want[key].dtype and np.array_equal(got, want[key]), (t, key)


def test_input_sample_device_rng_invariants(dev, oracle):
    from articulated_pose_amd.dataset import create_unit_data_batch, pack_cloud, tiled_size
    with np.load(G) as z:
        parts, N, nf, K, _perm, _want = load_case(z, "tile")
        parts2, _, nf2, K2, _, _ = load_case(z, "more")
    a = create_unit_data_batch([parts, parts2], N, [nf, nf2], 3, seed=5, device=dev)
    b = create_unit_data_batch([parts, parts2], N, [nf, nf2], 3, seed=5, device=dev)
    c = create_unit_data_batch([parts, parts2], N, [nf, nf2], 3, seed=6, device=dev)
    assert all(torch.equal(a[k], b[k]) for k in a) and not torch.equal(a["P"], c["P"])
    for i, (p, f) in enumerate(((parts, nf), (parts2, nf2))):
        raw = pack_cloud(p)
        P = a["P"][i].cpu().numpy()
        scaled = raw[:, :3] * np.float32(f)
        # every sampled row is a raw row (scaled), labels / masks agree with it
        key = {tuple(r): j for j, r in enumerate(scaled)}
        src = np.array([key[tuple(r)] for r in P])
        assert np.array_equal(a["cls_gt"][i].cpu().numpy(), raw[src, 3])
        assert np.array_equal(a["mask_array"][i].cpu().numpy().argmax(1), raw[src, 3].astype(int))
        assert np.array_equal(a["joint_cls_mask"][i].cpu().numpy(), (raw[src, 17] > 0).astype(np.float32))
        if raw.shape[0] >= N:
            assert len(set(src.tolist())) == N                   # a permutation: no row twice
        else:
            counts = np.bincount(src, minlength=raw.shape[0])
            assert counts.max() <= tiled_size(raw.shape[0], N) // raw.shape[0]   # at most tile_n copies of a raw row


def test_input_sample_argument_errors(dev):
    from articulated_pose_amd.dataset import create_unit_data_batch
    with pytest.raises(ValueError):
        create_unit_data_batch([], 1024, [], 3, device=dev)
    with pytest.raises(ValueError):
        create_unit_data_batch([np.zeros((0, 18), np.float32)], 1024, [1.0], 3, device=dev)
    with pytest.raises(RuntimeError):
        create_unit_data_batch([np.zeros((4, 18), np.float32)], 8, [1.0], 3, device="cpu")


def test_caller_supplied_perm_follows_numpy_index_rules(dev):
    """A caller's permutation is numpy's fancy index on the tiled cloud (lib/dataset.py:298-300): entries in [-size, -1] wrap,
    anything outside [-size, size) raises IndexError."""
    from articulated_pose_amd.dataset import create_unit_data_batch, tiled_size
    rng = np.random.RandomState(0)
    raw = rng.rand(10, 18).astype(np.float32)
    raw[:, 3] = rng.randint(0, 3, 10)
    N = 8
    size = tiled_size(10, N)
    perm = np.array([0, 3, -1, -size, size - 1, 5, -4, 2])
    out = create_unit_data_batch([raw], N, [1.0], 3, perms=[perm], device=dev)
    tiled = np.concatenate([raw] * (size // 10), axis=0)
    assert np.array_equal(out["P"][0].cpu().numpy(), tiled[perm][:, :3])
    for bad in (-size - 1, size):
        with pytest.raises(IndexError):
            create_unit_data_batch([raw], N, [1.0], 3, perms=[np.array([0, 1, 2, 3, 4, 5, 6, bad])], device=dev)


# 12 seeds in the suite; ANCSH_INPUT_SWEEP_SEEDS=N for a one-off long fuzz (profiles/r05_ops_fuzz.txt)
INPUT_SEEDS = range(int(os.environ.get("ANCSH_INPUT_SWEEP_SEEDS", "12")))


@pytest.mark.parametrize("seed", INPUT_SEEDS)
def test_input_sample_sweep(dev, seed):
    """Seeded sweep of ancsh_input_sample over ragged batches -- 1..5 clouds of 1..5 parts, raw clouds from a single point (tiled
    num_points + 1 times) to several times num_points, num_points 1..3000 -- against the oracle on the SAME numpy permutations: every
    output array bit for bit."""
    from articulated_pose_amd.dataset import create_unit_data_batch, tiled_size
    from oracle import input_oracle
    from golden.gen_input_golden import synthetic_parts
    rng = np.random.RandomState(8000 + seed)
    N = int([1, 2, 63, 64, 65, 1024, 2048][rng.randint(7)]) if seed % 2 else int(rng.randint(1, 3001))
    B = int(rng.randint(1, 6))
    clouds, nfs, perms, Ks = [], [], [], []
    for _ in range(B):
        K = int(rng.randint(1, 6))
        total = int([1, 2, N // 3 + 1, N, N + 1, 3 * N + 7][rng.randint(6)])
        cuts = np.sort(rng.randint(0, total + 1, K - 1)) if K > 1 else np.array([], int)
        sizes = np.diff(np.concatenate([[0], cuts, [total]])).astype(int)
        sizes = [int(x) for x in sizes]
        if sum(sizes) == 0:
            sizes[0] = 1
        clouds.append(synthetic_parts(rng, sizes))
        nfs.append(float(rng.uniform(0.2, 3.0)))
        Ks.append(K)
        perms.append(rng.permutation(tiled_size(sum(sizes), N)))
    Kmax = max(Ks)
    out = create_unit_data_batch(clouds, N, nfs, Kmax, perms=perms, device=dev)
    for b in range(B):
        want = input_oracle.create_unit_data(clouds[b], N, np.float32(nfs[b]), Kmax, perm=perms[b])
        for key in OUT_KEYS:
            got = out[key][b].cpu().numpy()
            assert got.dtype == want[key].dtype and np.array_equal(got, want[key]), (seed, b, key, N, Ks[b])
