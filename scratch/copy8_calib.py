#!/usr/bin/env python3
"""scratch/copy8_calib.py -- the known-traffic launches the FETCH_SIZE / WRITE_SIZE counters are calibrated on (run under
rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE): cagpu_debug_copy8 moves n float64 with the step kernels' access shape
(one 8-byte element per lane and instruction), i.e. reads exactly 8 n bytes and writes exactly 8 n.  Sizes: 40 MB (5 M
elements; the buffers are re-used, so the Infinity Cache holds them like it holds the simulator state between launches) and
400 MB (past the 256 MB Infinity Cache).  Also a torch copy of the same buffers (16 bytes per lane) for comparison."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_collision_avoidance_amd import _native as nat  # noqa: E402

lib = nat.lib()
dev = torch.device("cuda", 0)
for n in (5_000_000, 50_000_000):
    src = torch.rand((n,), dtype=torch.float64, device=dev)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(12):
        nat.check(lib.cagpu_debug_copy8(n, src.data_ptr(), dst.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    for _ in range(4):
        dst.copy_(src)
    torch.cuda.synchronize()
    print("copy8 n = %d: %d bytes read, %d bytes written per launch" % (n, 8 * n, 8 * n))
