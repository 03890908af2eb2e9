"""Read the reference's own real-data case (BASELINE.json configs[0]: the Cance catchment, 28 x 28 cells, 383 active,
3 gauges, 1440 hourly steps) from the DATA files under /root/reference/smash/dataset -- GeoTIFF forcing, CSV
discharge, the France flow-direction raster -- into the arrays the solver boundary takes.  Used only by
make_golden.py (fixture generation in the build container); nothing at test time reads /root/reference.

GDAL / h5py / pandas-side plumbing of the reference is restated with numpy + PIL:
  * mesh: smash/mesh/meshing.py:227-297 (_get_mesh_from_xy) + smash/mesh/mw_meshing.f90:7-110 (catchment_dln: the
    outlet is moved within +-max_depth cells to the cell whose upstream area matches the surveyed area best),
    gauges and areas of smash/dataset/Cance/mesh_Cance.py;
  * forcing: smash/core/_read_input_data.py:150-205 (_read_prcp: hourly tiles, x 0.1 mm) and :207-300 (_read_pet:
    daily inter-annual PET of the LEAP-year day with the same day-of-year, spread over the day with
    RATIO_PET_HOURLY, smash/core/_constant.py:47-75), windows by smash/tools/raster_handler.py:353-387;
  * observations: _read_input_data.py:25-80 (_read_qobs);
  * defaults of Model(): mwd_parameters.f90:150-167, mwd_states.f90:117-126, lr = dt * 5 / 3600 (_build_model.py:257).
The result is pinned by the discharge printed in the reference's docstring (smash/core/model.py:475-477), see
make_golden.py and tests/test_oracle_golden.py.
"""
import datetime as dt_
import glob
import os
import struct

import numpy as np

RATIO_PET_HOURLY = np.array([0, 0, 0, 0, 0, 0, 0, 0.035, 0.062, 0.079, 0.097, 0.11, 0.117, 0.117, 0.11, 0.097, 0.079,
                             0.062, 0.035, 0, 0, 0, 0, 0], dtype=np.float32)

GAUGES = dict(x=[840_261, 826_553, 828_269], y=[6_457_807, 6_467_115, 6_469_198],
              area=[381.7 * 1e6, 107 * 1e6, 25.3 * 1e6], code=["V3524010", "V3515010", "V3517010"])
START, END, DT = dt_.datetime(2014, 9, 15), dt_.datetime(2014, 11, 14), 3600


def read_tiff(path):
    """(array, xleft, ytop, xres, yres, nodata) of a single-band, strip-organised GeoTIFF; uncompressed strips are
    decoded here (PIL refuses 64-bit floats), anything else goes through PIL."""
    b = open(path, "rb").read()
    assert b[:4] == b"II*\x00", path
    off = struct.unpack("<I", b[4:8])[0]
    n = struct.unpack("<H", b[off:off + 2])[0]
    size = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 11: 4, 12: 8}
    fmt = {1: "B", 3: "H", 4: "I", 11: "f", 12: "d"}
    tags = {}
    for i in range(n):
        e = b[off + 2 + 12 * i: off + 14 + 12 * i]
        tag, typ, cnt = struct.unpack("<HHI", e[:8])
        sz = size[typ] * cnt
        data = e[8:8 + sz] if sz <= 4 else b[struct.unpack("<I", e[8:12])[0]:][:sz]
        tags[tag] = data if typ == 2 else struct.unpack("<%d%s" % (cnt, fmt[typ]), data) if typ != 5 else None
    w, h, bits, comp = tags[256][0], tags[257][0], tags[258][0], tags[259][0]
    sfmt = tags.get(339, (1,))[0]
    if comp == 1:
        raw = b"".join(b[o:o + c] for o, c in zip(tags[273], tags[279]))
        dtype = {(3, 64): "<f8", (3, 32): "<f4", (2, 32): "<i4", (1, 32): "<u4", (2, 16): "<i2", (1, 16): "<u2"}[(sfmt, bits)]
        arr = np.frombuffer(raw, dtype=dtype, count=w * h).reshape(h, w)
    else:
        from PIL import Image
        Image.MAX_IMAGE_PIXELS = None
        arr = np.array(Image.open(path))
    sx, sy = tags[33550][0], tags[33550][1]
    tp = tags[33922]
    nodata = float(tags[42113].split(b"\x00")[0]) if 42113 in tags else None
    return arr, tp[3] - tp[0] * sx, tp[4] + tp[1] * sy, sx, sy, nodata


def windowed(path, xmin, ymax, dx, nrow, ncol, lacuna=-99.0):
    """gdal_read_windowed_raster (raster_handler.py:43-95) for rasters already at the mesh resolution."""
    arr, xleft, ytop, xres, yres, nodata = read_tiff(path)
    assert xres == dx and yres == dx
    c0, r0 = int((xmin - xleft) / xres), int((ytop - ymax) / yres)
    sl = arr[r0:r0 + nrow, c0:c0 + ncol]
    out = sl.astype(np.float64)
    if nodata is not None:
        out[sl == nodata] = lacuna
    return out


def upstream_mask(ds, n, root):
    """Cells draining through `root` (mask_upstream_cells, mw_meshing.f90:7-45); ds = flat downstream index."""
    up = [[] for _ in range(n)]
    for c in np.flatnonzero(ds >= 0):
        up[ds[c]].append(int(c))
    return up


def build_mesh(dataset_root, max_depth=1):
    from smash_amd import synth
    flw, xmin, ymax, xres, yres, _ = read_tiff(os.path.join(dataset_root, "France_flwdir.tif"))
    flwdir = flw.astype(np.int32)
    nrow, ncol = flwdir.shape
    ds, _ = synth.downstream_index(flwdir)
    up = upstream_mask(ds, nrow * ncol, 0)

    def closure(root):
        seen, stack = [], [root]
        while stack:
            c = stack.pop()
            seen.append(c)
            stack.extend(up[c])
        return np.array(seen)

    x = np.array(GAUGES["x"], np.float32)
    y = np.array(GAUGES["y"], np.float32)
    area = np.array(GAUGES["area"], np.float32)
    mask = np.zeros(nrow * ncol, np.int32)
    rows, cols, area_dln = [], [], []
    for g in range(x.size):
        row, col = int((ymax - y[g]) / yres), int((x[g] - xmin) / xres)
        best = (np.float32(1.0), None, None)
        for i in range(-max_depth, max_depth + 1):            # column offset outermost (mw_meshing.f90:72-104)
            for j in range(-max_depth, max_depth + 1):
                r, c = row + j, col + i
                if 0 <= r < nrow and 0 <= c < ncol:
                    cells = closure(r * ncol + c)
                    tol = np.abs(area[g] - np.float32(cells.size) * np.float32(xres * yres)) / area[g]
                    if tol < best[0]:
                        best = (tol, (r, c), cells)
        (r, c), cells = best[1], best[2]
        rows.append(r); cols.append(c)
        area_dln.append(np.float32(cells.size * (xres * yres)))
        mask[cells] = 1
    mask = mask.reshape(nrow, ncol)
    rr, cc = np.flatnonzero(mask.any(axis=1)), np.flatnonzero(mask.any(axis=0))
    srow, erow, scol, ecol = rr[0], rr[-1] + 1, cc[0], cc[-1] + 1
    fd = flwdir[srow:erow, scol:ecol]
    act = mask[srow:erow, scol:ecol]
    flwacc = synth.flow_accumulation(fd)                       # over the bounding box, as the reference does
    path = synth.make_path(flwacc)
    gauge_pos = np.column_stack((np.array(rows) - srow, np.array(cols) - scol))
    m = synth.Mesh(fd.shape[0], fd.shape[1], xres, np.where(act == 1, fd, -99), np.where(act == 1, flwacc, -99), path, act, gauge_pos,
                   np.array(area_dln, np.float32))
    m.xmin, m.ymax = xmin + scol * xres, ymax - srow * yres
    m.code = list(GAUGES["code"])
    return m


def date_range():
    n = int((END - START).total_seconds() // DT)
    return [START + dt_.timedelta(seconds=DT * (k + 1)) for k in range(n)]


def read_forcing(cance_root, m):
    dates = date_range()
    nt = len(dates)
    prcp = np.full((m.nrow, m.ncol, nt), -99.0, np.float32, order="F")
    pet = np.full((m.nrow, m.ncol, nt), -99.0, np.float32, order="F")
    files = sorted(glob.glob(os.path.join(cance_root, "prcp", "**", "*tif*"), recursive=True))
    by_stamp = {}
    for f in files:
        by_stamp.setdefault(os.path.basename(f).split("_")[2], f)          # rain_precipitation_<stamp>_<stamp>.tif
    for k, d in enumerate(dates):
        f = by_stamp.get(d.strftime("%Y%m%d%H%M"))
        if f is not None:
            prcp[:, :, k] = windowed(f, m.xmin, m.ymax, m.dx, m.nrow, m.ncol) * 0.1      # prcp_conversion_factor
    # daily inter-annual PET: the file of the leap-year day with the same day-of-year (_read_input_data.py:222-287)
    pfiles = sorted(glob.glob(os.path.join(cance_root, "pet", "**", "*tif*"), recursive=True))
    leap0 = dt_.datetime(2020, 1, 1)
    for k, d in enumerate(dates):
        day = leap0 + dt_.timedelta(days=d.timetuple().tm_yday - 1)
        hit = [f for f in pfiles if day.strftime("%m%d") in f]
        if hit:
            pet[:, :, k] = windowed(hit[0], m.xmin, m.ymax, m.dx, m.nrow, m.ncol) * 1 * RATIO_PET_HOURLY[d.hour]
    return prcp, pet


def read_qobs(cance_root, m):
    nt = len(date_range())
    qobs = np.full((m.ng, nt), -99.0, np.float32)
    for i, code in enumerate(m.code):
        path = glob.glob(os.path.join(cance_root, "qobs", "**", f"*{code}*.csv"), recursive=True)
        assert len(path) == 1
        lines = open(path[0]).read().split("\n")
        header = dt_.datetime.strptime(lines[0].strip(), "%Y%m%d%H%M")
        time_diff = int((START - header).total_seconds() / DT) + 1
        vals = lines[1:]
        k = 0
        if time_diff > 0:
            src = vals[time_diff:]
        else:
            src, k = vals, -time_diff
        for line in src:
            if k >= nt:
                break
            try:
                qobs[i, k] = float(line)
            except ValueError:
                break
            k += 1
    return qobs


def default_fields(m):
    """Model() defaults (mwd_parameters.f90:150-167, mwd_states.f90:117-126, _build_model.py:257)."""
    from smash_amd import synth
    pv = dict(ci=1e-6, cp=200.0, beta=1000.0, cft=500.0, cst=500.0, alpha=0.9, exc=0.0, b=0.3, cusl1=100.0, cusl2=500.0,
              clsl=2000.0, ks=20.0, ds=0.02, dsm=0.33, ws=0.8, lr=DT * (5 / 3600))
    sv = dict(hi=0.01, hp=0.01, hft=0.01, hst=0.01, husl1=0.01, husl2=0.01, hlsl=0.01, hlr=0.000001)
    P = {k: np.full((m.nrow, m.ncol), pv[k], np.float32, order="F") for k in synth.PARAM_NAMES}
    S = {k: np.full((m.nrow, m.ncol), sv[k], np.float32, order="F") for k in synth.STATE_NAMES}
    return P, S


def load(reference_root="/root/reference"):
    ds_root = os.path.join(reference_root, "smash", "dataset")
    m = build_mesh(ds_root)
    prcp, pet = read_forcing(os.path.join(ds_root, "Cance"), m)
    qobs = read_qobs(os.path.join(ds_root, "Cance"), m)
    P, S = default_fields(m)
    return m, prcp, pet, qobs, P, S
