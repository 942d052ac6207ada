"""GPU parity of star_voxel_scatter through the C ABI: bit-exact against the oracle (numpy statements
of star/datasets/MultiTempDet.py:134-143 and their plain-C restatement), ragged / empty / duplicate / wrapping /
out-of-range inputs, both index widths, both output layouts, and the batch assembly feeding the model."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from oracle import c_oracle as OC
    from oracle import star_oracle as O
    from star_amd import ops
    from star_amd.datasets import voxel as V

DEV = "cuda"
DIMS = (256, 256, 13)


def _frames(rng, dims, counts, dtype=np.int32):
    return [np.stack([rng.randint(0, d, size=c) for d in dims], 1).astype(dtype) for c in counts]


def _dev(frames, dtype):
    off = torch.from_numpy(np.concatenate([[0], np.cumsum([len(f) for f in frames])]).astype(np.int64)).to(DEV)
    flat = np.concatenate([f.reshape(-1, 3) for f in frames] + [np.zeros((0, 3), dtype)], 0).astype(dtype)
    return torch.from_numpy(flat).to(DEV), off


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_scatter_bit_exact_ragged(dtype):
    rng = np.random.RandomState(0)
    frames = _frames(rng, DIMS, [25000, 0, 1, 3000, 0, 70000], dtype)
    frames[3][:50, 0] -= DIMS[0]                                   # numpy wraps negative indices
    frames[3][50:80, 2] -= DIMS[2]
    frames[0] = np.concatenate([frames[0], frames[0][:100]])       # duplicates
    idx, off = _dev(frames, dtype)
    got = ops.voxel_scatter(idx, off, DIMS).cpu().numpy()
    want = OC.voxel_dense(frames, DIMS)
    assert np.array_equal(got, want)
    assert np.array_equal(got[3], O.voxel_dense(frames[3], DIMS))  # and the reference's numpy statements themselves
    cf = ops.voxel_scatter(idx, off, DIMS, channels_first=True).cpu().numpy()
    assert np.array_equal(cf, want.transpose(0, 3, 1, 2))


def test_scatter_non_square_and_all_empty():
    rng = np.random.RandomState(1)
    dims = (32, 48, 5)
    frames = _frames(rng, dims, [300, 200, 7])
    idx, off = _dev(frames, np.int32)
    got = ops.voxel_scatter(idx, off, dims).cpu().numpy()
    assert got.shape == (3, 48, 32, 5) and np.array_equal(got, OC.voxel_dense(frames, dims))
    idx, off = _dev([np.zeros((0, 3), np.int32)] * 4, np.int32)
    out = torch.full((4, 48, 32, 5), 7.0, device=DEV)
    ops.voxel_scatter(idx, off, dims, out=out)
    assert float(out.abs().sum()) == 0.0                            # the whole output is written


def test_scatter_odd_dims_and_chunked_batches():
    rng = np.random.RandomState(2)
    dims = (30, 20, 3)
    frames = _frames(rng, dims, [100, 0, 555])
    idx, off = _dev(frames, np.int32)
    assert np.array_equal(ops.voxel_scatter(idx, off, dims).cpu().numpy(), OC.voxel_dense(frames, dims))
    cf = ops.voxel_scatter(idx, off, dims, channels_first=True).cpu().numpy()
    assert np.array_equal(cf, OC.voxel_dense(frames, dims).transpose(0, 3, 1, 2))
    # more frames than one cache-sized chunk holds (28 full-size frames per 96 MiB): the chunk loop, with empty chunks' worth
    # of frames in the middle, must give the same frames as the single-chunk call
    counts = [int(c) for c in rng.randint(0, 3000, size=70)]
    counts[20:50] = [0] * 30
    frames = _frames(rng, DIMS, counts)
    idx, off = _dev(frames, np.int32)
    one = ops.voxel_scatter(idx, off, DIMS)
    many = ops.voxel_scatter(idx, off, DIMS, host_offsets=off.cpu())
    assert torch.equal(one, many)
    assert np.array_equal(many[69].cpu().numpy(), O.voxel_dense(frames[69])) and np.array_equal(many[3].cpu().numpy(), O.voxel_dense(frames[3]))


def test_out_of_range_raises_like_numpy():
    dims = (32, 48, 5)
    for bad in ([[32, 0, 0]], [[0, 48, 0]], [[0, 0, 5]], [[-33, 0, 0]], [[0, 0, -6]]):
        frames = [np.array([[1, 2, 3]], np.int32), np.array(bad, np.int32)]
        idx, off = _dev(frames, np.int32)
        with pytest.raises(IndexError):
            ops.voxel_scatter(idx, off, dims)
        with pytest.raises(IndexError):
            O.voxel_dense(frames[1], dims)
        # without validation the good points still land and nothing is written out of bounds
        out = ops.voxel_scatter(idx, off, dims, validate=False).cpu().numpy()
        assert np.array_equal(out[0], O.voxel_dense(frames[0], dims)) and out[1].sum() == 0.0


def test_collate_feeds_the_model_layout():
    """Full-size batch: B=2 scenes x A=5 agents x T=3 (one empty agent slot), assembled by ONE scatter launch, equals
    the frames the reference's dataset + training loop would hand to CoModule (train_completion.py:241-275)."""
    rng = np.random.RandomState(5)
    B, A, T = 2, 5, 3
    scenes = []
    for b in range(B):
        sc = []
        for a in range(A):
            if b == 1 and a == 4:
                sc.append(None)
                continue
            vox = _frames(rng, DIMS, [rng.randint(5000, 30000) for _ in range(T)])
            sc.append({"voxels": vox, "teacher": np.concatenate(vox), "trans_matrices": rng.randn(A, 4, 4), "num_sensor": 5 - b})
        scenes.append(sc)
    data = V.collate_sparse(scenes, T, DEV)
    assert data["bev_seq"].shape == (A * B, 1, 256, 256, 13) and data["bev_seq_next"].shape == (A * B, T - 1, 256, 256, 13)
    assert data["bev_seq_teacher"].shape == (A * B, 1, 256, 256, 13) and data["trans_matrices"].shape == (B, A, A, 4, 4)
    bev, nxt, tea = data["bev_seq"].cpu().numpy(), data["bev_seq_next"].cpu().numpy(), data["bev_seq_teacher"].cpu().numpy()
    for a in range(A):
        for b in range(B):
            s, row = scenes[b][a], a * B + b
            if s is None:
                assert bev[row].sum() == 0 and nxt[row].sum() == 0 and tea[row].sum() == 0
                assert int(data["num_agent"][b, a]) == 0 and float(data["trans_matrices"][b, a].abs().sum()) == 0
                continue
            assert np.array_equal(bev[row, 0], O.voxel_dense(s["voxels"][0]))
            for t in range(1, T):
                assert np.array_equal(nxt[row, t - 1], O.voxel_dense(s["voxels"][t]))
            assert np.array_equal(tea[row, 0], O.voxel_dense(s["teacher"]))
            assert int(data["num_agent"][b, a]) == s["num_sensor"]
            assert np.array_equal(data["trans_matrices"][b, a].numpy(), s["trans_matrices"])


def test_dataset_files_to_batch(tmp_path):
    rng = np.random.RandomState(6)
    A, n, T = 3, 4, 2
    dims = DIMS
    scenes = [0, 0, 1, 1]
    files, truth = [], []
    for a in range(A):
        fa, ta = [], []
        for i in range(n):
            path = str(tmp_path / ("%d_%d.npy" % (a, i)))
            if a == 2 and i == 1:
                V.write_sample(path, None, None, None, 0)
                ta.append(None)
            else:
                idx = _frames(rng, dims, [2000])[0]
                V.write_sample(path, idx, idx[:500], rng.randn(A, 4, 4), A)
                ta.append(idx)
            fa.append(path)
        files.append(fa)
        truth.append(ta)
    ds = V.SparseSceneDataset(files, [scenes] * A, time_stamp=T)
    data = V.collate_sparse([ds[3], ds[2]], T, DEV)
    # frame 3 pairs with frame 2 (same scene); frame 2 is the start of its scene and pairs with itself
    assert np.array_equal(data["bev_seq_next"][0 * 2 + 0, 0].cpu().numpy(), O.voxel_dense(truth[0][2]))
    assert np.array_equal(data["bev_seq_next"][0 * 2 + 1, 0].cpu().numpy(), O.voxel_dense(truth[0][2]))
    assert np.array_equal(data["bev_seq"][2 * 2 + 1, 0].cpu().numpy(), O.voxel_dense(truth[2][2]))
    assert np.array_equal(data["bev_seq_teacher"][1 * 2 + 0, 0].cpu().numpy(), O.voxel_dense(truth[1][3][:500]))
