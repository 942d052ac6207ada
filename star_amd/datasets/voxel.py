"""Sparse-index side of the reference's V2X-Sim dataset classes (star/datasets/MultiTempDet.py,
MultiTempSeg.py), in front of the completion path.

The reference densifies every frame on the HOST (zeros -> fancy assignment -> rot90(k=3) -> float32,
MultiTempDet.py:134-143,201-209,244-247) and ships 3.4 MB per frame over PCIe.  Here only the index lists
cross (12 B per occupied voxel) and ONE star_voxel_scatter launch writes every frame of the batch -- current
frames, past frames and teacher frames -- straight into the layouts CoModule unpacks
(completion/train_completion.py:241-275): bev_seq [A*B,1,H,W,C], bev_seq_next [A*B,T-1,H,W,C] (or [A*B,0]),
bev_seq_teacher [A*B,1,H,W,C], agent-major rows a*B+b.

On-disk sample format (what the reference reads with np.load(allow_pickle=True).item(), MultiTempDet.py:39,98,
135,179-189,197-199): a dict with `voxel_indices_0` [n,3], `voxel_indices_teacher[_no_cross_road]`,
`trans_matrices[_no_cross_road]` [A,4,4], `num_sensor`, `target_agent_id`; an empty agent slot is the scalar 0.
The coperception base class (file discovery, detection targets) is a third-party package and not part of this
path; `SparseSceneDataset` keeps only what the completion loop consumes."""
import numpy as np
import torch

from .. import ops

DIMS = (256, 256, 13)     # coperception Config.map_dims for V2X-Sim: X, Y, Z voxels


def write_sample(path, voxel_indices, teacher_indices, trans_matrices, num_sensor, target_agent_id=0, no_cross_road=None):
    """One agent-frame file in the reference's format (None -> the empty-slot marker 0)."""
    if voxel_indices is None:
        np.save(path, 0)
        return
    d = {"voxel_indices_0": np.asarray(voxel_indices), "voxel_indices_teacher": np.asarray(teacher_indices),
         "trans_matrices": np.asarray(trans_matrices), "num_sensor": num_sensor, "target_agent_id": target_agent_id}
    if no_cross_road is not None:
        d["voxel_indices_teacher_no_cross_road"] = np.asarray(no_cross_road[0])
        d["trans_matrices_no_cross_road"] = np.asarray(no_cross_road[1])
    np.save(path, d, allow_pickle=True)


def read_sample(path):
    """-> dict, or None for an empty slot (the file holds 0; MultiTempDet.py:39-40)."""
    h = np.load(path, allow_pickle=True)
    if h.dtype != object:
        assert h == 0, "unexpected sample file %s" % path
        return None
    return h.item()


def past_frame_indices(seq_scenes, idx, time_stamp):
    """Frame numbers of the T-1 PAST frames paired with frame `idx` of one agent (MultiTempDet.py:228-233):
    walk back one frame at a time; a step that would leave the scene (or the sequence) is moved forward again
    until it is inside -- the start of a scene repeats its earliest usable frame."""
    out = []
    current = seq_scenes[idx]
    for k in range(idx - 1, idx - time_stamp, -1):
        while k < 0 or seq_scenes[k] != current:
            k += 1
        out.append(k)
    return out


def densify(frames, dims=DIMS, device="cuda", channels_first=False, validate=True):
    """frames: list of [n,3] integer arrays (n may be 0) -> float32 [F, Y, X, Z] on the device, one launch."""
    counts = [0 if f is None else len(f) for f in frames]
    offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64))
    nonempty = [np.asarray(f).reshape(-1, 3) for f in frames if f is not None and len(f)]
    if nonempty:
        cat = np.concatenate(nonempty, 0)
        assert cat.dtype.kind in "iu", "voxel indices must be integers, got %s" % cat.dtype
        fits32 = cat.dtype.itemsize <= 4 and cat.dtype != np.uint32 or (cat.min() >= -2**31 and cat.max() < 2**31)
        flat = torch.from_numpy(np.ascontiguousarray(cat.astype(np.int32 if fits32 else np.int64)))
    else:
        flat = torch.zeros(0, 3, dtype=torch.int32)
    pin = torch.cuda.is_available()
    idx = (flat.pin_memory() if pin else flat).to(device, non_blocking=True)
    off = (offsets.pin_memory() if pin else offsets).to(device, non_blocking=True)
    return ops.voxel_scatter(idx, off, dims, channels_first=channels_first, validate=validate, host_offsets=offsets)


def collate_sparse(scenes, time_stamp, device="cuda", dims=DIMS, no_cross_road=False, validate=True):
    """scenes: list (length B) of scenes; a scene is a list (length A) of per-agent dicts
         {"voxels": [idx_t0, idx_t-1, ...] (T arrays), "teacher": idx, "trans_matrices": [A,4,4], "num_sensor": n}
       or None for an empty slot (zero frames, num_sensor 0, zero poses: MultiTempDet.py:40-96).
    -> the dict completion/train_completion.py:256-275 assembles, dense frames already on the device."""
    B, A, T = len(scenes), len(scenes[0]), time_stamp
    cur, nxt, tea = [], [], []
    trans = np.zeros((B, A, A, 4, 4), dtype=np.float64)
    num = np.zeros((B, A), dtype=np.int64)
    for a in range(A):                      # agent-major: row a*B + b
        for b in range(B):
            s = scenes[b][a]
            if s is None:
                cur.append(None)
                tea.append(None)
                nxt.extend([None] * (T - 1))
                continue
            assert len(s["voxels"]) == T, "scene %d agent %d: %d frames for time_stamp %d" % (b, a, len(s["voxels"]), T)
            cur.append(s["voxels"][0])
            nxt.extend(s["voxels"][1:])
            tea.append(s["teacher"])
            trans[b, a] = s["trans_matrices"]
            num[b, a] = s["num_sensor"]
    if no_cross_road:
        num -= 1                             # train_completion.py:245-246
    AB = A * B
    X, Y, Z = dims
    dense = densify(cur + nxt + tea, dims, device, validate=validate)        # [AB*(T+1), Y, X, Z]
    data = {
        "bev_seq": dense[:AB].unsqueeze(1),
        "bev_seq_next": dense[AB:AB * T].view(AB, T - 1, Y, X, Z) if T > 1 else torch.zeros(AB, 0, device=device),
        "bev_seq_teacher": dense[AB * T:].unsqueeze(1),
        "trans_matrices": torch.from_numpy(trans),          # stays on the host (train_completion.py:265)
        "num_agent": torch.from_numpy(num).to(device),
    }
    return data


def discover(data_root, num_agent=6, no_cross_road=False):
    """File tables of a pre-processed V2X-Sim split laid out the way the reference's scripts pass it to the dataset class
    (completion/train_completion.py:66-69: `dataset_roots = [f"{data}/agent{i}" ...]`, agents 1.. when --no_cross_road):
    every agent directory holds one directory `<scene>_<frame>` per frame with the sample file inside.  The ordering --
    by (scene, frame) as integers -- and the scene id = the directory prefix are those of the coperception base class
    (V2XSimDet.__init__), a third-party package that is absent here: restated from its published source, parity unpinned.
    -> (files[a][i], seq_scenes[a][i]) for SparseSceneDataset."""
    import os
    files, seq_scenes = [], []
    for a in (range(1, num_agent) if no_cross_road else range(num_agent)):
        root = os.path.join(data_root, "agent%d" % a)
        dirs = [d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))]
        dirs.sort(key=lambda x: (int(x.split("_")[0]), int(x.split("_")[1])))
        seq_scenes.append([int(d.split("_")[0]) for d in dirs])
        row = []
        for d in dirs:
            inside = sorted(f for f in os.listdir(os.path.join(root, d)) if os.path.isfile(os.path.join(root, d, f)))
            assert inside, "no sample file in %s" % os.path.join(root, d)
            row.append(os.path.join(root, d, inside[0]))
        files.append(row)
    n = len(files[0])
    assert all(len(r) == n for r in files), "agents hold different numbers of frames: %r" % [len(r) for r in files]
    return files, seq_scenes


def _identity(scenes):     # module-level: DataLoader workers pickle the collate function
    return scenes


class SparseSceneDataset:
    """Per-agent frame files in the reference's format -> sparse scene samples for `collate_sparse`.
    files[a][i] is agent a's i-th frame file, seq_scenes[a][i] its scene id (the two tables the coperception
    base class builds, used at MultiTempDet.py:38,229-235)."""

    def __init__(self, files, seq_scenes, time_stamp=1, no_cross_road=False):
        self.files, self.seq_scenes, self.time_stamp, self.no_cross_road = files, seq_scenes, time_stamp, no_cross_road
        self.num_agent = len(files)

    def __len__(self):
        return len(self.files[0])

    def loader(self, batch_size, shuffle=True, num_workers=0, device="cuda", seed=None, rank=0, world=1):
        """Batches for CoModule: a torch DataLoader over the sparse samples (file reads in worker processes, nothing dense
        crosses PCIe), densified by one star_voxel_scatter launch per batch in the consuming process.  With world > 1 every
        rank walks its own share of one common permutation (whole scenes per rank), cut to equal length so that the ranks
        take the same number of steps.  Yields (data dict, scenes in this batch)."""
        for scenes in self.scene_batches(batch_size, shuffle, num_workers, seed, rank, world):
            yield collate_sparse(scenes, self.time_stamp, device=device, no_cross_road=self.no_cross_road), len(scenes)

    def scene_batches(self, batch_size, shuffle=True, num_workers=0, seed=None, rank=0, world=1):
        """The host half of `loader`: lists of sparse scenes, in this rank's order."""
        from torch.utils.data import DataLoader, Subset
        n = len(self)
        if shuffle:
            gen = torch.Generator()
            gen.manual_seed(0 if seed is None else int(seed))
            order = torch.randperm(n, generator=gen).tolist()
        else:
            order = list(range(n))
        if world > 1:
            order = order[:n - n % world][rank::world]
        yield from DataLoader(Subset(self, order), batch_size=batch_size, shuffle=False, num_workers=num_workers, collate_fn=_identity)

    def steps_per_epoch(self, batch_size, world=1):
        n = len(self)
        per_rank = (n - n % world) // world if world > 1 else n
        return (per_rank + batch_size - 1) // batch_size

    def __getitem__(self, idx):
        scene = []
        sfx = "_no_cross_road" if self.no_cross_road else ""
        for a in range(self.num_agent):
            d = read_sample(self.files[a][idx])
            if d is None:
                scene.append(None)
                continue
            vox = [d["voxel_indices_0"]]
            for k in past_frame_indices(self.seq_scenes[a], idx, self.time_stamp):
                dk = read_sample(self.files[a][k])
                vox.append(vox[0] if dk is None else dk["voxel_indices_0"])     # MultiTempDet.py:237-247
            scene.append({"voxels": vox, "teacher": d["voxel_indices_teacher" + sfx], "trans_matrices": d["trans_matrices" + sfx],
                          "num_sensor": d["num_sensor"]})
        return scene
