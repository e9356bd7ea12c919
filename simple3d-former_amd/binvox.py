"""`.binvox` input path: numpy-2-safe reader for the format utils/binvox_rw.py handles (header :73-89, run-length payload
:117-151, writer :246-290), producing the bit-packed grids the device unpacks (s3d_unpack_voxels).

File layout: ASCII header `#binvox 1 / dim d h w / translate x y z / scale s / data`, then (value, count) byte pairs in
x-z-y order (y fastest).  The reference decodes to a dense bool array and transposes to [x, y, z]
(read_as_3d_array, fix_coords=True); data/modelnet40.py:40 then ships np.int32 grids, 32 bits per voxel, to the GPU.  Here
the host only expands the runs straight into a [x, y, z] bit grid (1 bit per voxel, z fastest, LSB first) and the device
turns B packed grids into the fp32 [B,1,V,V,V] tensor the tokenizer reads: 32x less host->device traffic."""
import ctypes
import io

import numpy as np
import torch

from . import _lib as L


class BinvoxError(IOError):
    pass


def read_header(fp):
    """(dims, translate, scale) — same fields and same failure mode (IOError 'Not a binvox file') as binvox_rw.read_header."""
    line = fp.readline().strip()
    if not line.startswith(b'#binvox'):
        raise BinvoxError('Not a binvox file')
    dims = [int(t) for t in fp.readline().strip().split(b' ')[1:]]
    translate = [float(t) for t in fp.readline().strip().split(b' ')[1:]]
    scale = [float(t) for t in fp.readline().strip().split(b' ')[1:]][0]
    fp.readline()
    return dims, translate, scale


def read_dense(src, fix_coords=True):
    """Dense bool grid [x, y, z] (fix_coords) or [x, z, y] of a .binvox file / bytes object."""
    fp = io.BytesIO(src) if isinstance(src, (bytes, bytearray)) else src
    dims, translate, scale = read_header(fp)
    raw = np.frombuffer(fp.read(), dtype=np.uint8)
    if raw.size % 2:
        raise BinvoxError('truncated run-length payload')
    values, counts = raw[0::2], raw[1::2]
    n = int(dims[0]) * int(dims[1]) * int(dims[2])
    if int(counts.sum(dtype=np.int64)) != n:
        raise BinvoxError(f'run lengths cover {int(counts.sum(dtype=np.int64))} voxels, header says {n}')
    data = np.repeat(values.astype(bool), counts.astype(np.intp)).reshape(dims)
    if fix_coords:
        data = np.ascontiguousarray(np.transpose(data, (0, 2, 1)))
    return data, dims, translate, scale


def pack_grid(dense):
    """bool/0-1 grid (any shape, total size multiple of 32) -> uint32 words, 32 voxels per word, LSB first in memory order."""
    flat = np.ascontiguousarray(dense).reshape(-1).astype(bool)
    if flat.size % 32:
        raise ValueError('grid size must be a multiple of 32 voxels')
    return np.packbits(flat, bitorder='little').view('<u4')


def read_packed(src):
    """(uint32 words of the [x,y,z] grid, dims, translate, scale)."""
    dense, dims, translate, scale = read_dense(src, fix_coords=True)
    return pack_grid(dense), dims, translate, scale


def write(dense_xyz, fp, translate=(0.0, 0.0, 0.0), scale=1.0):
    """Writes a [x,y,z] occupancy grid as .binvox (runs of at most 255, x-z-y order) — the inverse of read_dense."""
    dense_xyz = np.asarray(dense_xyz).astype(bool)
    dims = (dense_xyz.shape[0], dense_xyz.shape[2], dense_xyz.shape[1])     # the reader reshapes the x-z-y payload by `dim`
    fp.write(b'#binvox 1\n')
    fp.write(('dim ' + ' '.join(str(d) for d in dims) + '\n').encode())
    fp.write(('translate ' + ' '.join(str(t) for t in translate) + '\n').encode())
    fp.write(('scale ' + str(scale) + '\n').encode())
    fp.write(b'data\n')
    flat = np.transpose(dense_xyz, (0, 2, 1)).reshape(-1)
    if flat.size == 0:
        return
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    starts = np.concatenate(([0], change))
    lengths = np.diff(np.concatenate((starts, [flat.size])))
    vals = flat[starts].astype(np.uint8)
    reps = (lengths + 254) // 255                                   # split long runs into <=255 pieces
    v = np.repeat(vals, reps)
    c = np.full(v.shape, 255, dtype=np.int64)
    last = np.cumsum(reps) - 1
    c[last] = lengths - (reps - 1) * 255
    out = np.empty(2 * v.size, dtype=np.uint8)
    out[0::2], out[1::2] = v, c.astype(np.uint8)
    fp.write(out.tobytes())


def unpack_to_device(words, batch, voxel_size, out=None, device='cuda'):
    """words: uint32 numpy array / int32 tensor holding `batch` packed V^3 grids -> fp32 device tensor [B,1,V,V,V]."""
    V = voxel_size
    nwords = batch * V * V * V // 32
    if isinstance(words, np.ndarray):
        words = torch.from_numpy(np.ascontiguousarray(words).view(np.int32).reshape(-1))
    if words.numel() != nwords:
        raise ValueError(f'expected {nwords} words for {batch} grids of {V}^3, got {words.numel()}')
    if not words.is_cuda:
        words = words.pin_memory().to(device, non_blocking=True) if torch.cuda.is_available() else words.to(device)
    if out is None:
        out = torch.empty(batch, 1, V, V, V, dtype=torch.float32, device=words.device)
    L.check(L.lib().s3d_unpack_voxels(L.ptr(words), L.ptr(out), ctypes.c_long(nwords), L.current_stream()), 'unpack_voxels')
    return out
