"""Training-time sample source: same class name, constructor argument and item layout as
dataloaders/ehem_dataset.py:8-66, so a training script written against the reference keeps working.

Files: the non-test output of the preprocessing (`<name>_<N>.npy`, int64 [N,4,6+] records, N in the file name).  An item is
one window of `context_size` consecutive records of one file: `data` int64 [c,4,3] = (level, octant, occ - 1), `pos` float32
[3,c] = the nodes' own positions normalised by the window's scalar min / max, `label` = the nodes' own occupancy - 1.
Windows of a file are visited in a random order drawn with `torch.randperm` when the file is (re)loaded; the file is chosen
by `index % n_files` (the reference's behaviour, including its statefulness across calls).
"""
import glob

import numpy as np
import torch
import torch.utils.data as data


class EHEMDataset(data.Dataset):
    def __init__(self, cfg):
        self.cfg = cfg
        self.file_names = sorted(glob.glob(cfg.root))
        assert self.file_names, "no file found!"
        self.total_point_num = sum(self._file_len(f) for f in self.file_names)
        self.root = cfg.root
        self.context_size = cfg.context_size
        self.max_time_each_file = 0          # windows in the current file
        self.cur_times = 0                   # windows of it already served
        self.cur_max_level = 0

    @staticmethod
    def _file_len(name):
        return int(name.split("_")[-1].split(".")[0])

    def __getitem__(self, index):
        c = self.context_size
        if self.cur_times >= self.max_time_each_file:        # current file exhausted (or first call): load the next one
            name = self.file_names[index % len(self.file_names)]
            self.cur_data = np.load(name)
            self.cur_data[:, :, 0] -= 1                      # occupancy 1..255 -> 0..254
            self.cur_max_level = max(self.cur_data[:, -1, 1])
            self.cur_times = 0
            self.max_time_each_file = self._file_len(name) // c
            self.order = torch.randperm(self.max_time_each_file)
        w = int(self.order[self.cur_times])
        rec = np.copy(self.cur_data[w * c:(w + 1) * c])

        def norm(p):
            lo, hi = p.min(), p.max()
            return ((p - lo) / (hi - lo)).astype(np.float32).transpose((1, 0))

        pos = norm(rec[:, -1, 3:6])
        xyz_pos = norm(rec[:, -1, 6:9]) if self.cfg.extra_pos else None
        d = rec[:, :, :3]
        d = np.concatenate((d[:, :, 1:], d[:, :, :1]), axis=2)      # (occ, level, octant) -> (level, octant, occ)
        label = np.copy(d[:, -1, 2])
        self.cur_times += 1
        return (d, pos, xyz_pos, label) if self.cfg.extra_pos else (d, pos, label)

    def __len__(self):
        return self.total_point_num // self.context_size
