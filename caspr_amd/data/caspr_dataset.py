"""Sequence loading for the CaSPR npz format (reference: caspr/data/caspr_dataset.py:148-208,277-343).

Host-side I/O only (numpy): one `frame_XXXXXXXX.npz` per time step with keys `nocs_data (n,3)`,
`depth_data (n,3)`, `obj_T (4,4)` (+ `rgb_data`, unused by the model).  `load_seq_path` and `select_item`
restate the reference's padding / time-stamping / sub-sampling rules so that data/demo-style directories
drive `caspr_amd.models.CaSPR` exactly as they drive the reference."""
import numpy as np
import torch

DEFAULT_MAX_TIMESTAMP = 5.0


def load_seq_path(seq_path_list, max_timestamp=DEFAULT_MAX_TIMESTAMP, expected_num_pts=4096):
    """caspr_dataset.py:148-208 -> nocs_seq (T,P,4) [xyz in the unit cube, t in [0,1]],
    depth_seq (T,P,4) [camera-frame xyz, t in [0,max_timestamp]], pose_seq (T,4,4); float64 like the reference."""
    seq_len = len(seq_path_list)
    step_size = 0.0 if seq_len == 1 else 1.0 / (seq_len - 1)
    nocs_seq = np.zeros((seq_len, expected_num_pts, 4))
    depth_seq = np.zeros((seq_len, expected_num_pts, 4))
    pose_seq = np.zeros((seq_len, 4, 4))
    for step_idx, pc_file in enumerate(seq_path_list):
        pc_data = np.load(pc_file)
        nocs_pc, depth_pc, pose = pc_data['nocs_data'], pc_data['depth_data'], pc_data['obj_T']
        if depth_pc.size == 0:      # warping-cars data has no depth: the NOCS cloud is the input (:173-175)
            depth_pc = nocs_pc
        if pose.size == 0:
            pose = np.zeros((4, 4))
        if np.count_nonzero(nocs_pc) == 0:   # blank frame: the reference stops filling here (:183-186)
            break
        if nocs_pc.shape[0] < expected_num_pts:   # pad by repeating the leading points (:188-195)
            pad_size = expected_num_pts - nocs_pc.shape[0]
            while pad_size > 0:
                nocs_pc = np.concatenate([nocs_pc, nocs_pc[:pad_size].reshape((-1, 3))], axis=0)
                depth_pc = np.concatenate([depth_pc, depth_pc[:pad_size].reshape((-1, 3))], axis=0)
                pad_size = expected_num_pts - nocs_pc.shape[0]
        pose_seq[step_idx] = pose
        t = np.ones((nocs_pc.shape[0], 1)) * step_size * step_idx
        nocs_seq[step_idx] = np.concatenate([nocs_pc, t], axis=1)
        t = max_timestamp * np.ones((depth_pc.shape[0], 1)) * step_size * step_idx
        depth_seq[step_idx] = np.concatenate([depth_pc, t], axis=1)
    return nocs_seq, depth_seq, pose_seq


def select_item(nocs_seq, depth_seq, seq_len, num_pts, steps=None, points=None, shift_time_to_zero=False):
    """The deterministic part of DynamicPCLDataset.__getitem__ (caspr_dataset.py:296-336): `steps` / `points`
    default to the first seq_len steps / first num_pts points (return_first_steps, random_point_sample=False,
    as test.py:112-115 configures the dataset).  -> (input (T,N,4) float32, output (T,N,4) float32)."""
    steps = sorted(np.arange(seq_len) if steps is None else steps)
    points = np.arange(num_pts) if points is None else points
    input_data = depth_seq[steps, :, :].copy()[:, points, :]
    output_data = nocs_seq[steps, :, :].copy()[:, points, :]
    if shift_time_to_zero:
        input_data[:, :, -1] -= np.min(input_data[:, :, -1])
        output_data[:, :, -1] -= np.min(output_data[:, :, -1])
    return torch.from_numpy(input_data.astype(np.float32)), torch.from_numpy(output_data.astype(np.float32))
