"""Drop-in for the reference's ``hsi_loader.HSIDataSet`` (hsi_loader.py:5-133): same constructor and
item tuples -- (XP f32[C,w,w], X f32[bands], Y int) for 'label' / 'unlabel' / 'test', (XP, X) for
'wholeset' -- read from the ``.npy`` files ``sample_generation.py`` writes.  ``SyntheticHSIDataSet``
serves N(0,1) patches of any window shape when the datasets are not on disk (they are not shipped).
``device_arrays()`` hands the whole split to the GPU once: the training driver keeps it resident in
HBM and gathers batches by index there instead of copying 24.6 MB over PCIe per step."""
import numpy as np
import torch
from torch.utils import data

_ROOTS = {1: './dataset/PaviaU/', 2: './dataset/Salinas/', 3: './dataset/Houston/', 4: './dataset/Indian_pines/'}
_SCENES = {1: (610, 340), 2: (512, 217), 3: (349, 1905), 4: (145, 145)}       # rows, cols of the scenes train.py:75-90 names


def _tile_to(arr, max_iters):
    reps, rem = divmod(int(max_iters), len(arr))
    parts = [arr] * reps + ([arr[:rem]] if rem else [])
    return np.concatenate(parts) if parts else arr[:0]


class HSIDataSet(data.Dataset):
    def __init__(self, dataID, setindex='label', max_iters=None, num_unlabel=1000, root=None):
        self.setindex = setindex
        self.root = root or _ROOTS[int(dataID)]
        XP = np.load(self.root + 'XP.npy', mmap_mode='r')
        X = np.load(self.root + 'X.npy', mmap_mode='r')
        Y = np.load(self.root + 'Y.npy') - 1
        if setindex == 'wholeset':
            self.XP, self.X, self.Y = XP, X, None
            return
        fname = {'label': 'train_array.npy', 'unlabel': 'unlabel_array.npy', 'test': 'test_array.npy'}[setindex]
        idx = np.load(self.root + fname)
        if setindex == 'unlabel':
            idx = idx[:num_unlabel]
        self.XP, self.X, self.Y = np.asarray(XP[idx]), np.asarray(X[idx]), Y[idx]
        if max_iters is not None and setindex in ('label', 'unlabel'):
            self.XP, self.X, self.Y = (_tile_to(a, max_iters) for a in (self.XP, self.X, self.Y))

    def __len__(self):
        return len(self.X)

    def __getitem__(self, index):
        XP = np.array(self.XP[index], dtype=np.float32)
        X = np.array(self.X[index], dtype=np.float32)
        if self.Y is None:
            return XP, X
        return XP, X, int(self.Y[index])

    def device_arrays(self, device):
        XP = torch.from_numpy(np.ascontiguousarray(self.XP, dtype=np.float32)).to(device)
        X = torch.from_numpy(np.ascontiguousarray(self.X, dtype=np.float32)).to(device)
        Y = None if self.Y is None else torch.from_numpy(np.asarray(self.Y, dtype=np.int64)).to(device)
        return XP, X, Y

    def cube_source(self, device, scene=None, dataID=None):
        """The 'wholeset' as the scene it was cut from, for whole-image inference without the materialised patches
        (tools.hyper_tools.test_whole): ``cube.npy`` ([rows, cols, C], the z-scored / PCA'd scene the patches were cut
        from -- INTEGRATION.md says which line of the reference's preprocessing has it in hand) next to XP.npy.  None
        when it is not there (the caller then streams the materialised patches, as the reference does): the cube is
        NOT rebuilt from XP.npy -- that gather touches every page of a ~20 GB file to recover 0.25 % of it."""
        import os
        from cmlpl_amd.infer import CubeSource
        if self.setindex != 'wholeset':
            raise ValueError("cube_source() is for the 'wholeset'")
        path = self.root + 'cube.npy'
        if not os.path.exists(path):
            return None
        cube = np.load(path, mmap_mode='r')
        if scene is None and dataID is not None:
            scene = _SCENES.get(int(dataID))
        if cube.ndim != 3 or (scene is not None and tuple(cube.shape[:2]) != tuple(scene)):
            return None
        cube = torch.from_numpy(np.ascontiguousarray(cube, dtype=np.float32)).to(device)
        X = torch.from_numpy(np.ascontiguousarray(self.X, dtype=np.float32)).to(device)
        if cube.shape[0] * cube.shape[1] != X.shape[0]:
            return None
        return CubeSource(cube, X)


class SyntheticHSIDataSet(data.Dataset):
    """Seeded stand-in with the same item tuples; class-dependent mean so that training has signal."""

    def __init__(self, shape, length, setindex='label', seed=1088, separable=1.0):
        C, H, W, bands, K = shape
        g = torch.Generator().manual_seed(seed)
        proto_g = torch.Generator().manual_seed(4242)
        self.Y = torch.randint(0, K, (length,), generator=g)
        proto_p = torch.randn(K, C, 1, 1, generator=proto_g) * separable
        proto_x = torch.randn(K, bands, generator=proto_g) * separable
        self.XP = torch.randn(length, C, H, W, generator=g) + proto_p[self.Y]
        self.X = torch.randn(length, bands, generator=g) + proto_x[self.Y]
        self.setindex = setindex

    def __len__(self):
        return len(self.X)

    def __getitem__(self, index):
        if self.setindex == 'wholeset':
            return self.XP[index].numpy(), self.X[index].numpy()
        return self.XP[index].numpy(), self.X[index].numpy(), int(self.Y[index])

    def device_arrays(self, device):
        return self.XP.to(device), self.X.to(device), self.Y.to(device)


class SyntheticScene:
    """A seeded synthetic scene for whole-image inference: a cube [rows, cols, C] and spectra [rows * cols, bands] whose
    pixels carry the class-dependent means of ``SyntheticHSIDataSet`` (same prototypes), plus the per-pixel labels.
    The classes lie in square REGIONS one and a half windows wide (a real scene's fields and roofs; each region its own
    seeded class), so that a window is mostly one class, like the training patches, which carry one class mean over the
    whole window -- with a class per pixel the spatial branch met windows it never trained on."""

    def __init__(self, shape, rows, cols, seed=3, separable=1.0):
        C, H, W, bands, K = shape
        g = torch.Generator().manual_seed(seed)
        proto_g = torch.Generator().manual_seed(4242)
        self.rows, self.cols, self.window = int(rows), int(cols), H
        blk = max(1, (3 * H) // 2)
        nbr, nbc = (rows + blk - 1) // blk, (cols + blk - 1) // blk
        region = torch.randint(0, K, (nbr, nbc), generator=g)
        self.Y = region.repeat_interleave(blk, 0)[:rows].repeat_interleave(blk, 1)[:, :cols].reshape(-1).contiguous()
        proto_p = torch.randn(K, C, 1, 1, generator=proto_g) * separable
        proto_x = torch.randn(K, bands, generator=proto_g) * separable
        self.cube = (torch.randn(rows * cols, C, generator=g) + proto_p[self.Y].view(-1, C)).view(rows, cols, C)
        self.X = torch.randn(rows * cols, bands, generator=g) + proto_x[self.Y]

    def __len__(self):
        return self.rows * self.cols

    def cube_source(self, device):
        from cmlpl_amd.infer import CubeSource
        return CubeSource(self.cube.to(device).contiguous(), self.X.to(device).contiguous())
