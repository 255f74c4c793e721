"""Seeded, counter-based synthetic data for parity tests, goldens and bench.

No dataset or checkpoint ships with the reference, so every
parity/throughput input is produced here from integer seeds:

* ``fill_module_``      deterministic weights keyed by parameter *name* (so the
                        reference module tree and this repo's mirror get
                        identical tensors without storing them in fixtures);
* ``bev_tile``          WHU-Lane-shaped pre-rasterised BEV tile;
* ``las_points``        LAS-shaped point records for the rasteriser.

Only integer hashing, adds and multiplies are used (no libm calls), so values are
bit-reproducible across hosts.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on a uint64 array."""
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def fnv1a64(s):
    h = 0xCBF29CE484222325
    for b in s.encode('utf-8'):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def u64_stream(seed, n, stream=0):
    base = _splitmix64(np.array([(int(seed) * 0x9E3779B97F4A7C15 + int(stream) * 0xD1B54A32D192ED03)
                                 & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
    ctr = np.arange(n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        return _splitmix64(ctr * np.uint64(0x2545F4914F6CDD1D) + base)


def uniform(seed, n, stream=0):
    """U[0,1) float64 with 53 random bits."""
    return (u64_stream(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normalish(seed, n, stream=0):
    """Approximately N(0,1): Irwin-Hall sum of 4 uniforms, variance-normalised (no libm)."""
    acc = np.zeros(n, dtype=np.float64)
    for k in range(4):
        acc += uniform(seed, n, stream * 4 + k + 1000)
    return (acc - 2.0) * 1.7320508075688772  # var of sum = 4/12


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
def _fill(t, vals):
    import torch
    with torch.no_grad():
        t.copy_(torch.from_numpy(np.ascontiguousarray(vals.reshape(tuple(t.shape)))).to(t.dtype))


def fill_module_(module, seed=2021, prefix=''):
    """Overwrite every parameter/buffer of ``module`` with name-keyed synthetic values.

    Rules (by owning module type): conv/linear weight ~ N(0, 1/fan_in), bias ~ N(0, .05²);
    BatchNorm gamma ~ U(.8,1.2), beta ~ N(0,.05²), running_mean ~ N(0,.05²),
    running_var ~ U(.6,1.4); Group/LayerNorm gamma ~ U(.8,1.2), beta ~ N(0,.05²);
    bare Parameters (pos_embedding, emb_*) ~ N(0, .1²).
    """
    import torch.nn as nn
    owners = {}
    for mname, m in module.named_modules():
        for pname, _ in list(m.named_parameters(recurse=False)) + list(m.named_buffers(recurse=False)):
            owners[(mname + '.' if mname else '') + pname] = (m, pname)
    sd = module.state_dict()
    for name, t in sd.items():
        m, pname = owners.get(name, (None, name))
        key = fnv1a64(prefix + name) ^ (int(seed) * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
        n = t.numel()
        if pname == 'num_batches_tracked':
            t.zero_()
            continue
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.GroupNorm, nn.LayerNorm)):
            if pname == 'weight':
                v = 0.8 + 0.4 * uniform(key, n)
            elif pname == 'running_var':
                v = 0.6 + 0.8 * uniform(key, n)
            else:  # bias, running_mean
                v = 0.05 * normalish(key, n)
        elif isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Linear)):
            if pname == 'weight':
                fan_in = int(np.prod(t.shape[1:]))
                v = normalish(key, n) * (1.0 / np.sqrt(fan_in))
            else:
                v = 0.05 * normalish(key, n)
        elif pname == 'weight' and t.dim() == 5:      # spconv layout [kD,kH,kW,Cin,Cout]: He-normal over the window
            v = normalish(key, n) * np.sqrt(2.0 / int(np.prod(t.shape[:4])))
        else:
            v = 0.1 * normalish(key, n)
        _fill(t, v.astype(np.float32))
    return module


def lidar_points(seed, n=4194304, extent=57.6):
    """[N,4] float32 ego-frame points for the sparse-conv path: the `las_points` cloud centred on
    the sensor (x, y in +-extent/2, road sheet ~1.6 m below it), intensity normalised like `read_las`
    ((clip(i,800,33000)-800)/33000).  The cloud is larger than the [-15,15]x[-25,25]x[-2,2] crop on purpose."""
    p = las_points(seed, n, extent).astype(np.float64)
    out = np.empty_like(p)
    out[:, 0] = p[:, 0] - extent / 2
    out[:, 1] = p[:, 1] - extent / 2
    out[:, 2] = p[:, 2] - 2.2
    out[:, 3] = (np.clip(p[:, 3], 800.0, 33000.0) - 800.0) / 33000.0
    return out.astype(np.float32)


# --------------------------------------------------------------------------------------
# pre-rasterised BEV tiles
# --------------------------------------------------------------------------------------
def bev_tile_u8(seed, size=1152):
    """uint8 HWC tile: background U(0,.15), 6 planted near-vertical stripes 3 px wide
    (alternating solid / dashed 40-on-40-off) at intensity U(.6,1), G = smooth plane."""
    H = W = size
    n = H * W
    r = (uniform(seed, n, 1) * 0.15).reshape(H, W)
    b = (uniform(seed, n, 2) * 0.15).reshape(H, W)
    par = uniform(seed, 64, 3)
    rows = np.arange(H, dtype=np.float64)[:, None]
    cols = np.arange(W, dtype=np.float64)[None, :]
    g = 0.3 + 0.2 * (rows / H) * (par[0] - 0.5) + 0.2 * (cols / W) * (par[1] - 0.5)
    g = np.broadcast_to(g, (H, W)).copy()
    stripe_int = (0.6 + 0.4 * uniform(seed, n, 4)).reshape(H, W)
    for k in range(6):
        x0 = (0.12 + 0.152 * k + 0.03 * (par[4 + k] - 0.5)) * W
        slope = 0.12 * (par[12 + k] - 0.5)
        centre = x0 + slope * rows                      # [H,1]
        on = np.abs(cols - centre) <= 1.0               # 3 px wide
        if k % 2 == 1:                                  # dashed 40-on-40-off
            on = on & (((rows.astype(np.int64) + int(par[20 + k] * 80)) // 40) % 2 == 0)
        r = np.where(on, stripe_int, r)
        b = np.where(on, stripe_int, b)
    img = np.stack([r, g, b], axis=2)
    return np.clip(np.floor(img * 255.0 + 0.5), 0, 255).astype(np.uint8)


def bev_tile(seed, size=1152):
    """float32 CHW tile in [0,1] = u8/255 — the `load_img` contract
    (reference datasets/laserlane_proposals.py:85-98)."""
    u8 = bev_tile_u8(seed, size)
    return (u8.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1).copy()


def bev_batch(seeds, size=1152):
    return np.stack([bev_tile(s, size) for s in seeds], axis=0)


# --------------------------------------------------------------------------------------
# LAS-shaped points
# --------------------------------------------------------------------------------------
def las_points(seed, n=4194304, extent=57.6):
    """[N,4] float32 records {x,y,z,raw_intensity} in the tile-local frame.

    70 % uniform over the extent×extent tile, 30 % on 6 lane stripes; z = plane + noise;
    raw intensity (u16 domain, stored as f32): road U(800,9000), stripes U(15000,33000).
    Points are in acquisition order, i.e. NOT spatially sorted."""
    u = uniform(seed, n, 11)
    v = uniform(seed, n, 12)
    sel = uniform(seed, n, 13)
    lane = (uniform(seed, n, 14) * 6).astype(np.int64)
    par = uniform(seed, 32, 15)
    on_stripe = sel >= 0.7
    x = u * extent                                           # along image rows
    y_road = v * extent
    x0 = (0.12 + 0.152 * lane + 0.03 * (par[lane] - 0.5)) * extent
    slope = 0.12 * (par[8 + lane] - 0.5)
    y_lane = x0 + slope * x + (v - 0.5) * 0.15               # 15 cm wide paint
    y = np.where(on_stripe, y_lane, y_road)
    y = np.clip(y, 0.0, extent - 1e-3)
    z = 0.02 * x + 0.01 * y + 0.03 * normalish(seed, n, 16)
    inten = np.where(on_stripe, 15000.0 + 18000.0 * uniform(seed, n, 17),
                     800.0 + 8200.0 * uniform(seed, n, 18))
    inten = np.floor(inten)
    return np.stack([x, y, z, inten], axis=1).astype(np.float32)


def las_point_records(points, scale=1e-3):
    """[N,4] float32 {x, y, z, raw intensity} -> the N point-data records of an ASPRS LAS 1.2 file, point format 0 (20 bytes: X Y Z as
    int32 = round(coordinate / scale) with offset 0, intensity u16, return / classification / scan-angle / user-data bytes, point source
    id u16), as one uint8 array - what a LAS reader hands to lm_las_decode_points."""
    n = points.shape[0]
    rec = np.zeros(n, dtype=np.dtype([('X', '<i4'), ('Y', '<i4'), ('Z', '<i4'), ('intensity', '<u2'), ('flags', 'u1'), ('cls', 'u1'),
                                      ('angle', 'i1'), ('user', 'u1'), ('src', '<u2')]))
    assert rec.dtype.itemsize == 20
    for k, name in enumerate(('X', 'Y', 'Z')):
        rec[name] = np.rint(points[:, k].astype(np.float64) / scale).astype(np.int32)
    rec['intensity'] = np.clip(points[:, 3], 0, 65535).astype(np.uint16)
    rec['flags'] = 0x11            # return 1 of 1
    rec['cls'] = 2                 # ground
    return rec.view(np.uint8)


def apply_gains_(module, gains):
    """Multiply named parameters in place: gains = {state-dict key: factor}.  Used to derive better-conditioned synthetic heads
    from the seeded weights (e.g. a regression layer whose outputs stay inside one bin) - same keys in the reference and here."""
    import torch
    sd = module.state_dict()
    with torch.no_grad():
        for k, g in gains.items():
            sd[k].mul_(float(g))
    return module
