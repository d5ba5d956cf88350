"""Device-resident window dataset + the reference's batch ordering.

Replaces ``BatchDictDataset`` / ``_H5BatchIterableDataset``
(/root/reference/deepof/clustering/dataset.py:29-670): instead of materialising every window to
HDF5 and re-reading it each epoch through h5py + H2D copies, the windows (or, better, the
un-windowed frame tables) are uploaded to HBM once (288 GB/GPU) and every batch is produced on
device by ``dof_window_gather``.

The batch order is the reference's arithmetic, reproduced exactly (dataset.py:589-622):
batch starts ``arange(0, n, bs)``; per-epoch ``numpy.random.default_rng((seed + epoch) % 2**32)``
shuffle of the starts (block shuffle, windows inside a batch stay contiguous); truncate to a
multiple of the world size; rank r takes ``starts[r::world]``; the last batch may be ragged
(``drop_last=False``).
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import _capi


def reorder_and_reshape(data: np.ndarray) -> np.ndarray:
    """(n, W, 3N) column blocks [x|y|s] -> (n, W, N, 3)   (dataset.py:16-26)."""
    assert data.shape[2] % 3 == 0, "Error! Number of columns is not a multiple of 3 (x, y, speed)!"
    n = data.shape[2] // 3
    return np.stack([data[:, :, 0:n], data[:, :, n:2 * n], data[:, :, 2 * n:3 * n]], axis=-1)


def video_ranges(video_idx: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(starts, ends) of the contiguous per-video runs of the concatenated window list (dataset.py:486-502)."""
    vid = np.asarray(video_idx)
    if vid.shape[0] == 0:
        return np.array([], dtype=np.int64), np.array([], dtype=np.int64)
    change = np.flatnonzero(vid[1:] != vid[:-1]) + 1
    bounds = np.concatenate(([0], change, [vid.shape[0]])).astype(np.int64)
    return bounds[:-1], bounds[1:]


def block_bootstrap_starts(rng: np.random.Generator, video_idx: np.ndarray, batch_size: int, target_batches: int,
                           block_len: int = 250) -> np.ndarray:
    """Block bootstrap of batch starts (dataset.py:505-559): draw a video, then a block of ceil(block_len / bs)
    consecutive full batches inside it, until ``target_batches`` starts are collected (sampling with replacement).
    Consumes ``rng`` exactly as the reference does, so the same seed yields the same batches."""
    v_starts, v_ends = video_ranges(video_idx)
    v_lens = (v_ends - v_starts).astype(np.int64)
    ok = v_lens >= batch_size
    v_starts, v_ends, v_lens = v_starts[ok], v_ends[ok], v_lens[ok]
    if len(v_lens) == 0:
        raise RuntimeError("No video segment long enough to provide a full batch.")
    per_block = int(max(1, np.ceil(int(block_len) / batch_size)))
    out = np.empty((target_batches,), dtype=np.int64)
    filled = 0
    while filled < target_batches:
        v = int(rng.choice(len(v_lens), p=None))
        vs, ve, m = int(v_starts[v]), int(v_ends[v]), int(v_lens[v])
        n_take = min(per_block, m // batch_size)
        last_start = ve - n_take * batch_size
        if last_start < vs:
            continue
        s0 = int(rng.integers(vs, last_start + 1))
        for j in range(n_take):
            if filled >= target_batches:
                break
            out[filled] = s0 + j * batch_size
            filled += 1
    return out


def batch_starts(n_samples: int, batch_size: int, epoch: int, seed: Optional[int], shuffle: bool,
                 world_size: int = 1, rank: int = 0, drop_last: bool = False, video_idx: Optional[np.ndarray] = None,
                 bootstrap: bool = False, bootstrap_block_len: int = 250) -> np.ndarray:
    """Start indices of this rank's batches for 1-based ``epoch`` (the reference's ``_epoch`` counter);
    dataset.py:585-618: seeded block shuffle, truncation to a multiple of the world size, optional block
    bootstrap (same generator, after the shuffle), then the rank's strided share."""
    if drop_last:
        starts = np.arange(0, (n_samples // batch_size) * batch_size, batch_size, dtype=np.int64)
    else:
        starts = np.arange(0, n_samples, batch_size, dtype=np.int64)
    base_seed = seed if seed is not None else 0
    rng = np.random.default_rng((base_seed + epoch) % (2 ** 32))
    if shuffle:
        rng.shuffle(starts)
    if world_size > 1:
        starts = starts[: (len(starts) // world_size) * world_size]
    if bootstrap:
        if video_idx is None:
            raise ValueError("bootstrap needs the per-window video index")
        starts = block_bootstrap_starts(rng, video_idx, batch_size, len(starts), bootstrap_block_len)
    if world_size > 1:
        starts = starts[rank::world_size]
    return starts


def n_batches(n_samples: int, batch_size: int, world_size: int = 1, drop_last: bool = False) -> int:
    """len(loader) of the reference (dataset.py:469-484)."""
    total = (n_samples // batch_size) if drop_last else ((n_samples + batch_size - 1) // batch_size)
    if world_size > 1:
        total = (total // world_size) * world_size // world_size
    return total


def _frame_table_of(windows: np.ndarray, stride: Optional[int] = None, chunk: int = 8192):
    """(table, stride) if ``windows`` (n, W, C) are the stride-s sliding windows of one (frames, C) table -- every element
    of every window is compared with the window before it (NaN == NaN), no sampling -- else None.  ``stride``: the step
    to test (default: the smallest s in 1 .. W for which windows 0 and 1 overlap)."""
    n, W = windows.shape[0], windows.shape[1]
    if n == 0 or windows.ndim != 3:
        return None
    if n == 1:
        return windows[0], (stride or 1)

    def same(p, q):
        # bit patterns: windows cut from one table are copies, so identical bits is the exact test (NaNs included) and one
        # integer compare per element instead of array_equal(equal_nan=True)'s five passes; a difference in bits only
        # (-0.0 against 0.0) falls back to the value comparison
        if p.dtype == np.float32 and p.strides[-1] == 4 and q.strides[-1] == 4:
            if np.array_equal(p.view(np.uint32), q.view(np.uint32)):
                return True
        return bool(np.array_equal(p, q, equal_nan=True)) if p.dtype.kind == "f" else bool(np.array_equal(p, q))

    if stride is None:
        stride = next((s for s in range(1, W + 1) if s == W or same(windows[1, : W - s], windows[0, s:])), None)
    if stride is None or stride > W:
        return None
    if stride < W:
        for lo in range(1, n, chunk):
            hi = min(n, lo + chunk)
            if not same(windows[lo:hi, : W - stride], windows[lo - 1:hi - 1, stride:]):
                return None
    tail = windows[1:, W - min(stride, W):, :].reshape(-1, windows.shape[2])
    return np.concatenate([windows[0], tail], axis=0), stride


class WindowDataset:
    """All windows of a ``{video_key: (nodes (n,W,3N), edges (n,W,E), angles)}`` dict, resident on ``device``.

    Stored in the reference's batch layout x (n,W,N,3), a (n,W,E,1) fp32 (what the HDF5 files held).
    ``from_tables`` keeps only the frame tables and gathers windows on the fly (W-fold less memory).
    """

    def __init__(self, device):
        self.device = torch.device(device)
        self.x: Optional[torch.Tensor] = None
        self.a: Optional[torch.Tensor] = None
        self.node_table: Optional[torch.Tensor] = None
        self.edge_table: Optional[torch.Tensor] = None
        self.row_start: Optional[torch.Tensor] = None
        self.video_idx: Optional[np.ndarray] = None
        self.keys: List[str] = []
        self.length = 0
        self.x_shape: Tuple[int, int, int] = (0, 0, 0)
        self.a_shape: Tuple[int, int, int] = (0, 0, 0)
        self._epoch = 0
        self._lib = None

    # ---- construction ----------------------------------------------------------------------
    @classmethod
    def from_preprocessed(cls, preprocessed: Dict, device, lib=None) -> "WindowDataset":
        """The reference's input format: ``{video: (node windows (n,W,3N), edge windows (n,W,E)[, angle windows])}``.

        With ``lib`` (the HIP library) the W-fold redundant windows are NOT uploaded: they are stride-s sliding windows
        over a frame table (``rolling_window``, /root/reference/deepof/utils.py:3354-3377), so the table is rebuilt
        from them on the host -- window 0 plus the last s rows of every later window -- after checking, exactly and
        over every element, that the windows really overlap that way (``_frame_table_of``), and batches come from
        ``dof_window_gather`` like those of ``from_tables`` (bit-identical to the materialised form; 1/W of the bytes
        cross PCIe and stay in HBM).  If any video is not a regular sliding-window set (shuffled windows, a hand-made
        array) the materialised form below is used."""
        ds = cls(device)
        keys = list(preprocessed.keys())
        angs = []
        for key in keys:
            if len(preprocessed[key]) > 2 and preprocessed[key][2] is not None:
                angs.append(np.asarray(preprocessed[key][2], dtype=np.float32))
        # angle windows (n, W, A): host-resident, only the teacher's optional angle view reads them (dataset.py:81-92)
        ds.angles = np.concatenate(angs) if len(angs) == len(keys) and angs and angs[0].shape[-1] > 0 else None
        if lib is not None:
            rebuilt = []
            for key in keys:
                nodes, edges = np.asarray(preprocessed[key][0]), np.asarray(preprocessed[key][1])
                nt = _frame_table_of(nodes)
                et = _frame_table_of(edges, stride=None if nt is None else nt[1]) if nt is not None else None
                if nt is None or et is None or nt[1] != et[1]:
                    rebuilt = None
                    break
                rebuilt.append((nt[0], et[0], nt[1], nodes.shape[0], nodes.shape[1]))
            if rebuilt is not None and len({r[4] for r in rebuilt}) == 1:
                ds._lib = lib
                starts, vid, off = [], [], 0
                for i, (nt, et, stride, n, _w) in enumerate(rebuilt):
                    starts.append(off + np.arange(n, dtype=np.int64) * stride)
                    vid.append(np.full(n, i, dtype=np.int32))
                    off += nt.shape[0]
                W = rebuilt[0][4]
                ds.node_table = torch.from_numpy(np.concatenate([r[0] for r in rebuilt]).astype(np.float32)).to(ds.device)
                ds.edge_table = torch.from_numpy(np.concatenate([r[1] for r in rebuilt]).astype(np.float32)).to(ds.device)
                ds.row_start = torch.from_numpy(np.concatenate(starts)).to(ds.device)
                ds.video_idx = np.concatenate(vid)
                ds.keys = keys
                ds.length = int(ds.row_start.numel())
                ds.x_shape = (W, ds.node_table.shape[1] // 3, 3)
                ds.a_shape = (W, ds.edge_table.shape[1], 1)
                return ds
        xs, as_, vid = [], [], []
        for i, key in enumerate(keys):
            nodes, edges = preprocessed[key][0], preprocessed[key][1]
            nodes, edges = np.asarray(nodes), np.asarray(edges)
            xs.append(reorder_and_reshape(nodes).astype(np.float32))
            as_.append(np.expand_dims(edges, -1).astype(np.float32))
            vid.append(np.full(nodes.shape[0], i, dtype=np.int32))
            ds.keys.append(key)
        x, a = np.concatenate(xs), np.concatenate(as_)
        ds.x = torch.from_numpy(x).to(ds.device)
        ds.a = torch.from_numpy(a).to(ds.device)
        ds.video_idx = np.concatenate(vid)
        ds.length = x.shape[0]
        ds.x_shape, ds.a_shape = tuple(x.shape[1:]), tuple(a.shape[1:])
        return ds

    @classmethod
    def from_indexed(cls, dataset, device) -> "WindowDataset":
        """An indexed in-memory dataset in the reference's item format -- ``dataset[i] = (x (T,N,3), a (T,E,1), [angles,]
        idx, vid)`` with ``x_shape`` / ``a_shape`` attributes (BatchDictDataset, /root/reference/deepof/clustering/dataset.py:16-181;
        the stand-in of the reference's tests, tests/test_build_models.py:43-103) -- materialised once on ``device`` in index
        order.  ``vid`` (the last element of an item) becomes the video index of the window."""
        ds = cls(device)
        n = len(dataset)
        items = [dataset[i] for i in range(n)]
        ds.x = torch.stack([torch.as_tensor(it[0], dtype=torch.float32) for it in items]).contiguous().to(ds.device)
        ds.a = torch.stack([torch.as_tensor(it[1], dtype=torch.float32) for it in items]).contiguous().to(ds.device)
        ds.video_idx = np.asarray([int(it[-1]) for it in items], dtype=np.int32) if n and len(items[0]) >= 4 else \
            np.zeros(n, dtype=np.int32)
        ds.keys = [f"video{v}" for v in sorted(set(ds.video_idx.tolist()))]
        ds.angles = None
        ds.length = n
        ds.x_shape = tuple(int(v) for v in getattr(dataset, "x_shape", tuple(ds.x.shape[1:])))
        ds.a_shape = tuple(int(v) for v in getattr(dataset, "a_shape", tuple(ds.a.shape[1:])))
        return ds

    @classmethod
    def from_tables(cls, tables: Dict, window_size: int, window_step: int, device, lib) -> "WindowDataset":
        """``tables``: {video_key: (node_table (frames,3N), edge_table (frames,E))}, un-windowed."""
        ds = cls(device)
        ds._lib = lib
        nts, ets, starts, vid, off = [], [], [], [], 0
        for i, key in enumerate(tables.keys()):
            nt, et = (np.asarray(t, dtype=np.float32) for t in tables[key][:2])
            nw = (nt.shape[0] - window_size) // window_step + 1
            starts.append(off + np.arange(nw, dtype=np.int64) * window_step)
            vid.append(np.full(nw, i, dtype=np.int32))
            nts.append(nt)
            ets.append(et)
            off += nt.shape[0]
            ds.keys.append(key)
        ds.node_table = torch.from_numpy(np.concatenate(nts)).to(ds.device)
        ds.edge_table = torch.from_numpy(np.concatenate(ets)).to(ds.device)
        ds.row_start = torch.from_numpy(np.concatenate(starts)).to(ds.device)
        ds.video_idx = np.concatenate(vid)
        ds.length = int(ds.row_start.numel())
        n, e = ds.node_table.shape[1] // 3, ds.edge_table.shape[1]
        ds.x_shape, ds.a_shape = (window_size, n, 3), (window_size, e, 1)
        return ds

    @classmethod
    def from_device_tables(cls, pre, window_size: int, window_step: int, lib, keys: Optional[List[str]] = None) -> "WindowDataset":
        """Windows over the frame tables ``deepof_amd.preprocess.preprocess_tables`` left on the device (nothing is
        copied): stride-``window_step`` windows inside every video, never across two (extract_windows,
        /root/reference/deepof/utils.py:3380-3474).  ``keys`` selects videos (e.g. the training or the test ones)."""
        ds = cls(pre.node_table.device)
        ds._lib = lib
        starts, vid = [], []
        for i, key in enumerate(pre.keys):
            if keys is not None and key not in keys:
                continue
            lo, hi = int(pre.video_off[i]), int(pre.video_off[i + 1])
            nw = (hi - lo - window_size) // window_step + 1
            if nw <= 0:
                continue
            starts.append(lo + np.arange(nw, dtype=np.int64) * window_step)
            vid.append(np.full(nw, len(ds.keys), dtype=np.int32))
            ds.keys.append(key)
        if not starts:
            raise ValueError("no video is long enough for one window")
        ds.node_table, ds.edge_table = pre.node_table, pre.edge_table
        ds.row_start = torch.from_numpy(np.concatenate(starts)).to(ds.device)
        ds.video_idx = np.concatenate(vid)
        ds.length = int(ds.row_start.numel())
        ds.x_shape = (window_size, pre.node_table.shape[1] // 3, 3)
        ds.a_shape = (window_size, pre.edge_table.shape[1], 1)
        ds.angles = None
        return ds

    def __len__(self):
        return self.length

    # ---- batches ---------------------------------------------------------------------------
    def fetch(self, s: int, e: int, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
              ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Windows [s, e) as contiguous device tensors x (b,W,N,3), a (b,W,E,1).  ``out`` = (x, a) buffers of that
        shape to fill in place (the static batch buffers of a captured step)."""
        if self.x is not None:
            if out is None:
                return self.x[s:e], self.a[s:e]
            out[0].copy_(self.x[s:e])
            out[1].copy_(self.a[s:e])
            return out
        W, N, _ = self.x_shape
        E = self.a_shape[1]
        b = e - s
        if out is None:
            out = (torch.empty(b, W, N, 3, device=self.device), torch.empty(b, W, E, 1, device=self.device))
        x, a = out
        assert tuple(x.shape) == (b, W, N, 3) and tuple(a.shape) == (b, W, E, 1) and x.is_contiguous() and a.is_contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
        rows = self.row_start[s:e]  # a slice of the contiguous start-row list: no copy
        if getattr(self, "window_storage", "fp32") == "bf16":
            # BASELINE's bf16 configuration: the batch is STORED as bf16 (half the bytes written by the gather and read
            # back by the step); the step's fp32 kernels read an exactly widened copy.  Values = the fp32 batch rounded
            # to nearest-even bf16.
            xb, ab = self.fetch_bf16(s, e, self._bf16_scratch(b))
            _capi.check(self._lib, self._lib.dof_widen_bf16(xb.data_ptr(), x.data_ptr(), xb.numel(), stream), "dof_widen_bf16")
            _capi.check(self._lib, self._lib.dof_widen_bf16(ab.data_ptr(), a.data_ptr(), ab.numel(), stream), "dof_widen_bf16")
            return x, a
        _capi.check(self._lib, self._lib.dof_window_gather(self.node_table.data_ptr(), self.edge_table.data_ptr(),
                                                           rows.data_ptr(), b, W, N, E, x.data_ptr(), a.data_ptr(),
                                                           stream), "dof_window_gather")
        return x, a

    def _bf16_scratch(self, b: int):
        cache = self.__dict__.setdefault("_bf16_buffers", {})
        if b not in cache:
            W, N, _ = self.x_shape
            E = self.a_shape[1]
            cache[b] = (torch.empty(b, W, N, 3, dtype=torch.bfloat16, device=self.device),
                        torch.empty(b, W, E, 1, dtype=torch.bfloat16, device=self.device))
        return cache[b]

    def fetch_bf16(self, s: int, e: int, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Windows [s, e) as bf16 tensors x (b,W,N,3), a (b,W,E,1) written by the gather itself (``dof_window_gather_bf16``:
        3,024 instead of 5,824 bytes per C2 window).  Frame-table datasets only."""
        if self.x is not None:
            raise ValueError("fetch_bf16 needs a frame-table dataset (from_tables / from_preprocessed / from_device_tables)")
        W, N, _ = self.x_shape
        E = self.a_shape[1]
        b = e - s
        if out is None:
            out = (torch.empty(b, W, N, 3, dtype=torch.bfloat16, device=self.device),
                   torch.empty(b, W, E, 1, dtype=torch.bfloat16, device=self.device))
        x, a = out
        assert x.dtype == torch.bfloat16 and a.dtype == torch.bfloat16 and tuple(x.shape) == (b, W, N, 3)
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
        rows = self.row_start[s:e]
        _capi.check(self._lib, self._lib.dof_window_gather_bf16(self.node_table.data_ptr(), self.edge_table.data_ptr(),
                                                                rows.data_ptr(), 0, 0, b, W, N, E, x.data_ptr(), a.data_ptr(),
                                                                stream), "dof_window_gather_bf16")
        return x, a

    def iter_ranges(self, batch_size: int, shuffle: bool, seed: Optional[int], world_size: int = 1, rank: int = 0,
                    drop_last: bool = False) -> Iterator[Tuple[int, int]]:
        """One epoch of this rank's batches as window ranges [s, e) in the reference loader's order (every batch is a
        contiguous run of windows: the loader shuffles batch STARTS only, dataset.py:589-634)."""
        self._epoch += 1
        boot = bool(getattr(self, "bootstrap_training", False)) and shuffle
        for s in batch_starts(self.length, batch_size, self._epoch, seed, shuffle, world_size, rank, drop_last,
                              self.video_idx, boot, getattr(self, "bootstrap_block_len", 250)):
            s = int(s)
            yield s, min(s + batch_size, self.length)

    def iter_batches(self, batch_size: int, shuffle: bool, seed: Optional[int], world_size: int = 1, rank: int = 0,
                     drop_last: bool = False) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, np.ndarray]]:
        """One epoch: yields (x, a, idx[int64 device], video_idx[int32 host]) like the reference loader."""
        for s, e in self.iter_ranges(batch_size, shuffle, seed, world_size, rank, drop_last):
            x, a = self.fetch(s, e)
            idx = torch.arange(s, e, device=self.device, dtype=torch.int64)
            yield x, a, idx, self.video_idx[s:e]
