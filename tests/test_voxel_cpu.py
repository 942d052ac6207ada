"""CPU checks for the dataset densification row: the plain-C oracle against the reference's own
numpy statements, the past-frame lookup against its statement-for-statement restatement, and the sample file
format round trip.  No HIP call."""
import numpy as np
import pytest

from oracle import c_oracle as OC
from oracle import star_oracle as O
from star_amd.datasets import voxel as V


def _frames(rng, n_frames, dims, counts):
    return [np.stack([rng.randint(0, d, size=c) for d in dims], 1).astype(np.int64) for c in counts[:n_frames]]


def test_c_oracle_matches_numpy_statements():
    rng = np.random.RandomState(0)
    dims = (256, 256, 13)
    frames = _frames(rng, 4, dims, [25000, 0, 1, 3000])
    frames[3][:50, 0] -= dims[0]      # negative indices wrap (numpy semantics)
    frames[3][50:80, 2] -= dims[2]
    frames[0] = np.concatenate([frames[0], frames[0][:100]])   # duplicates
    got = OC.voxel_dense(frames, dims)
    for f, idx in enumerate(frames):
        assert np.array_equal(got[f], O.voxel_dense(idx, dims)), f
    # the orientation: voxel (x, y, z) lands on [y, X-1-x, z]
    one = OC.voxel_dense([np.array([[3, 10, 5]])], dims)[0]
    assert one[10, 255 - 3, 5] == 1.0 and one.sum() == 1.0


def test_non_square_dims_and_errors():
    rng = np.random.RandomState(1)
    dims = (32, 48, 5)
    frames = _frames(rng, 2, dims, [300, 200])
    got = OC.voxel_dense(frames, dims)
    assert got.shape == (2, 48, 32, 5)
    for f, idx in enumerate(frames):
        assert np.array_equal(got[f], O.voxel_dense(idx, dims))
    bad = np.array([[32, 0, 0]])
    with pytest.raises(IndexError):
        O.voxel_dense(bad, dims)
    with pytest.raises(IndexError):
        OC.voxel_dense([bad], dims)
    with pytest.raises(IndexError):
        OC.voxel_dense([np.array([[0, -49, 0]])], dims)


def test_past_frame_lookup_matches_reference_statements():
    rng = np.random.RandomState(2)
    for _ in range(50):
        n = rng.randint(1, 30)
        scenes = np.sort(rng.randint(0, 4, size=n)).tolist()
        for T in (1, 2, 3, 5):
            for idx in range(n):
                assert V.past_frame_indices(scenes, idx, T) == O.past_frames(scenes, idx, T)
    assert V.past_frame_indices([7, 7, 7, 9, 9], 3, 3) == [3, 3]      # start of a scene repeats itself
    assert V.past_frame_indices([7, 7, 7, 9, 9], 2, 3) == [1, 0]
    assert V.past_frame_indices([7, 7, 7], 0, 3) == [0, 0]


def test_sample_file_round_trip(tmp_path):
    rng = np.random.RandomState(3)
    idx = _frames(rng, 1, (256, 256, 13), [100])[0]
    tm = rng.randn(5, 4, 4)
    V.write_sample(str(tmp_path / "a.npy"), idx, idx[:40], tm, 4, target_agent_id=2, no_cross_road=(idx[:10], tm * 2))
    V.write_sample(str(tmp_path / "empty.npy"), None, None, None, 0)
    d = V.read_sample(str(tmp_path / "a.npy"))
    assert np.array_equal(d["voxel_indices_0"], idx) and d["num_sensor"] == 4 and d["target_agent_id"] == 2
    assert np.array_equal(d["trans_matrices_no_cross_road"], tm * 2)
    assert V.read_sample(str(tmp_path / "empty.npy")) is None
