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


def _write_split(root, n_agents, frames, rng, empty=()):
    """frames: list of (scene, frame) -> <root>/agent<a>/<scene>_<frame>/0.npy in the reference's sample format."""
    import os
    truth = {}
    for a in range(n_agents):
        for (sc, fr) in frames:
            d = os.path.join(root, "agent%d" % a, "%d_%d" % (sc, fr))
            os.makedirs(d)
            if (a, sc, fr) in empty:
                V.write_sample(os.path.join(d, "0.npy"), None, None, None, 0)
                continue
            vox = np.stack([rng.randint(0, 256, 200), rng.randint(0, 256, 200), rng.randint(0, 13, 200)], 1)
            tea = np.stack([rng.randint(0, 256, 300), rng.randint(0, 256, 300), rng.randint(0, 13, 300)], 1)
            trans = rng.randn(n_agents, 4, 4)
            V.write_sample(os.path.join(d, "0.npy"), vox, tea, trans, n_agents, a)
            truth[(a, sc, fr)] = (vox, tea, trans)
    return truth


def test_discover_orders_frames_by_scene_then_frame_and_dataset_reads_them(tmp_path):
    rng = np.random.RandomState(3)
    frames = [(10, 2), (9, 15), (2, 10), (2, 9), (10, 0)]           # string order would differ from integer order
    truth = _write_split(str(tmp_path), 3, frames, rng, empty={(1, 9, 15)})
    files, seq_scenes = V.discover(str(tmp_path), num_agent=3)
    want = sorted(frames)
    assert seq_scenes == [[s for s, _ in want]] * 3
    for a in range(3):
        assert [f.split("/")[-2] for f in files[a]] == ["%d_%d" % sf for sf in want]
    ds = V.SparseSceneDataset(files, seq_scenes, time_stamp=2)
    assert len(ds) == 5 and ds.steps_per_epoch(2) == 3 and ds.steps_per_epoch(2, world=2) == 1
    for i, (sc, fr) in enumerate(want):
        scene = ds[i]
        for a in range(3):
            if (a, sc, fr) not in truth:
                assert scene[a] is None
                continue
            vox, tea, trans = truth[(a, sc, fr)]
            assert np.array_equal(scene[a]["voxels"][0], vox) and np.array_equal(scene[a]["teacher"], tea)
            assert np.array_equal(scene[a]["trans_matrices"], trans) and scene[a]["num_sensor"] == 3
            k = V.past_frame_indices(seq_scenes[a], i, 2)[0]
            past = truth.get((a,) + want[k])
            assert np.array_equal(scene[a]["voxels"][1], vox if past is None else past[0])     # empty past slot -> the current frame
    # --no_cross_road: agents 1.. only
    files1, _ = V.discover(str(tmp_path), num_agent=3, no_cross_road=True)
    assert len(files1) == 2 and "agent1" in files1[0][0]
    # rank shares: disjoint, equal length, together one permutation prefix
    a0 = [id_ for b in ds.scene_batches(1, shuffle=True, seed=7, rank=0, world=2) for id_ in map(id, b)]
    seen = []
    for r in range(2):
        for b in ds.scene_batches(2, shuffle=True, seed=7, rank=r, world=2):
            seen.append((r, len(b)))
    assert [n for _, n in seen] == [2, 2] and len(a0) == 2
