#!/usr/bin/env python3
"""Fixture generator (build container only; nothing at test or bench time reads /root/reference): the reference's own 1-km D8
flow-direction raster of France -- the data file its mesh generator reads (smash/mesh/meshing.py:216-297, dataset
smash/dataset/France_flwdir.tif) -- stored as data: tests/golden/mesh/france_d8.npz = {flwdir int8 (nrow, ncol), 0 = no data; dx; the
raster's upper-left corner}.  smash_amd.synth.make_mesh_france() builds meshes (whole raster, or its largest basins) from it by the
rules tests/golden/cance_io.py restates; bench.py --mesh france:* times the routing schedule on that real river network."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main(reference_root="/root/reference"):
    import cance_io
    flw, xmin, ymax, xres, yres, nodata = cance_io.read_tiff(os.path.join(reference_root, "smash", "dataset", "France_flwdir.tif"))
    assert xres == yres == 1000.0
    fd = np.where((flw >= 1) & (flw <= 8), flw, 0).astype(np.int8)
    out = os.path.join(HERE, "mesh", "france_d8.npz")
    np.savez_compressed(out, flwdir=fd, dx=np.float32(xres), xmin=np.float64(xmin), ymax=np.float64(ymax))
    print(out, fd.shape, int((fd > 0).sum()), "cells with a direction,", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main(*sys.argv[1:])
