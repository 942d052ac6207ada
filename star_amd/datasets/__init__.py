from .voxel import SparseSceneDataset, collate_sparse, densify, discover, past_frame_indices, read_sample, write_sample  # noqa: F401
