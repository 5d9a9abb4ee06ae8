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


# ---------------------------------------------------------------------------------------------
# Dataset object (reference: caspr_dataset.py:22-146, 211-349): same constructor arguments, same item tuple
# ((input (T,N,4), output (T,N,4)), [pose (T,4,4)], model_id, seq_id), same use of numpy's global RNG (so a seeded run
# draws the same steps / points as the reference).  Errors raise instead of exit().
# ---------------------------------------------------------------------------------------------
import glob
import os

from torch.utils.data import Dataset

DEFAULT_EXPECTED_SEQ_LEN = 10
DEFAULT_EXPECTED_NUM_PTS = 4096
# ShapeNet car ids whose renders are blank spheres (caspr_dataset.py:10-13)
BAD_MODELS = ('93ce8e230939dfc230714334794526d4', '207e69af994efa9330714334794526d4', '2307b51ca7e4a03d30714334794526d4')


def parse_dataset_cfg(cfg_file_path):
    """Dataset .cfg = whitespace-separated `--flag value...` lines (data/configs/*.cfg; caspr_dataset.py:26-34)."""
    opts = {"data": None, "splits": None, "max_timestamp": DEFAULT_MAX_TIMESTAMP, "expected_num_pts": DEFAULT_EXPECTED_NUM_PTS,
            "expected_seq_len": DEFAULT_EXPECTED_SEQ_LEN}
    kinds = {"data": list, "splits": list, "max_timestamp": float, "expected_num_pts": int, "expected_seq_len": int}
    with open(cfg_file_path) as f:
        tokens = f.read().split()
    key = None
    for tok in tokens:
        if tok.startswith("--"):
            key = tok[2:].replace("-", "_")
            if key not in kinds:
                raise ValueError("unknown dataset option %s in %s" % (tok, cfg_file_path))
            if kinds[key] is list:
                opts[key] = []
        elif key is None:
            raise ValueError("value %r without an option in %s" % (tok, cfg_file_path))
        elif kinds[key] is list:
            opts[key].append(tok)
        else:
            opts[key] = kinds[key](tok)
    if not opts["data"]:
        raise ValueError("%s: --data is required" % cfg_file_path)

    class Cfg:
        pass
    cfg = Cfg()
    cfg.__dict__.update(opts)
    return cfg


def _visible_dirs(root):
    return [p for p in (os.path.join(root, f) for f in sorted(os.listdir(root)) if not f.startswith('.')) if os.path.isdir(p)]


def load_time_data(data_roots, split, train_frac, val_frac, splits_dirs=None, data_seq_len=DEFAULT_EXPECTED_SEQ_LEN, log=print):
    """Frame-file lists of every sequence of `split` (caspr_dataset.py:36-146).  Layout: root/model_id/seq_id/*frame*.npz.
    With split files (`<dir>/<split>_split.txt`, one model id per line) the listed models are taken; otherwise the
    sorted models are cut into the leading train_frac, the next val_frac and the rest (test)."""
    all_paths = []
    for src, root in enumerate(data_roots):
        if not os.path.exists(root):
            raise FileNotFoundError('Could not find %s!' % root)
        wanted = None
        if splits_dirs is not None:
            split_file = os.path.join(splits_dirs[src], split + '_split.txt')
            if not os.path.exists(split_file):
                raise FileNotFoundError('There is no split file for the requested split! (%s)' % split_file)
            with open(split_file) as f:
                wanted = [m for m in f.read().split('\n') if m != '']
        model_dirs = _visible_dirs(root) if wanted is None else [os.path.join(root, m) for m in wanted]
        per_model = []
        for mdir in model_dirs:
            model_id = mdir.split('/')[-1]
            if wanted is not None and not os.path.exists(mdir):
                log('WARNING: Could not find model %s requested in the split file! Skipping...' % model_id)
                continue
            if model_id in BAD_MODELS:
                continue
            seqs = []
            for sdir in _visible_dirs(mdir):
                frames = sorted(glob.glob(os.path.join(sdir, '*frame*.npz')))
                if len(frames) == data_seq_len:
                    seqs.append(frames)
                else:
                    log('Found %d frames at %s...skipping!' % (len(frames), sdir))
            per_model.append(seqs)
        n_models = len(per_model)
        if splits_dirs is None:
            if train_frac + val_frac > 1.0:
                raise ValueError('Training and validation fraction must be less than 1.0!')
            n_train = int(train_frac * n_models)
            n_val = int(val_frac * n_models)
            bounds = {"train": (0, n_train), "val": (n_train, n_train + n_val), "test": (n_train + n_val, n_models)}
            lo, hi = bounds[split]
            per_model = per_model[lo:hi]
        for seqs in per_model:
            all_paths.extend(seqs)
    return all_paths


class DynamicPCLDataset(Dataset):
    """Point-cloud sequences for T-NOCS regression / reconstruction training (caspr_dataset.py:211-349)."""

    def __init__(self, data_cfg, split='train', train_frac=0.8, val_frac=0.1, num_pts=1024, seq_len=5, shift_time_to_zero=False,
                 random_point_sample=True, random_point_sample_per_step=False):
        if split not in ('train', 'test', 'val'):
            raise ValueError('Split %s is not a valid option. Choose train, test, or val.' % split)
        cfg = parse_dataset_cfg(data_cfg)
        self.data_paths, self.split_paths = cfg.data, cfg.splits
        self.data_seq_len, self.expected_num_pts, self.max_timestamp = cfg.expected_seq_len, cfg.expected_num_pts, cfg.max_timestamp
        self.split, self.train_frac, self.val_frac = split, train_frac, val_frac
        self.num_pts, self.seq_len = num_pts, seq_len
        self.shift_time_to_zero = shift_time_to_zero
        self.random_point_sample = random_point_sample
        self.random_point_sample_per_step = random_point_sample_per_step
        self.return_pose_data = False
        self.return_first_steps = False
        self.seq_data_paths = load_time_data(self.data_paths, split, train_frac, val_frac, self.split_paths, data_seq_len=self.data_seq_len)
        self.data_len = len(self.seq_data_paths)

    def __len__(self):
        return self.data_len

    def set_return_pose_data(self, return_pose):
        self.return_pose_data = return_pose

    def set_return_first_steps(self, return_first_steps):
        self.return_first_steps = return_first_steps

    def __getitem__(self, idx):
        files = self.seq_data_paths[idx]
        model_id, seq_id = files[0].split('/')[-3], files[0].split('/')[-2]
        nocs, depth, pose = load_seq_path(files, max_timestamp=self.max_timestamp, expected_num_pts=self.expected_num_pts)
        # RNG draws in the reference's order: steps first, then points (caspr_dataset.py:293-310)
        steps = np.arange(self.seq_len) if self.return_first_steps else np.random.choice(nocs.shape[0], self.seq_len, replace=False)
        steps = sorted(steps)
        if self.random_point_sample:
            pts = np.random.choice(nocs.shape[1], self.num_pts, replace=False)
        elif self.random_point_sample_per_step:
            pts = np.stack([np.random.choice(nocs.shape[1], self.num_pts, replace=False) for _ in range(nocs.shape[0])], axis=0)
        else:
            pts = np.arange(self.num_pts)
        if self.random_point_sample or not self.random_point_sample_per_step:
            inp, out = depth[steps][:, pts, :], nocs[steps][:, pts, :]
        else:
            # one independent point subset per step; as in the reference the subsets are indexed by their own position
            # (row k of `pts` goes with the k-th SAMPLED step), which requires seq_len == the data's sequence length
            t_idx = np.repeat(np.arange(pts.shape[0]), pts.shape[1])
            inp = depth[steps][t_idx, pts.reshape(-1), :].reshape(pts.shape[0], pts.shape[1], -1)
            out = nocs[steps][t_idx, pts.reshape(-1), :].reshape(pts.shape[0], pts.shape[1], -1)
        inp, out = inp.copy(), out.copy()
        if self.shift_time_to_zero:
            inp[:, :, -1] -= np.min(inp[:, :, -1])
            out[:, :, -1] -= np.min(out[:, :, -1])
        item = [(torch.from_numpy(inp.astype(np.float32)), torch.from_numpy(out.astype(np.float32)))]
        if self.return_pose_data:
            item.append(pose[steps, :])
        item.extend([model_id, seq_id])
        return tuple(item)
