"""Row-group scan (cm_scan_cl_fwd, xdbl mode) at small batches: unchunked against time_chunks = 2 .. 16.
   python tools/bench_scan_chunks.py     (GPU box; prints one line per (shape, chunks))"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_asr_amd import ops, _native  # noqa: E402


def case(b, l, e, rank, chunk_list, reps=20):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    P = 16 if rank <= 16 else 32
    z = rnd(b, l, e).bfloat16()
    dirs = []
    ycat = torch.empty(b, l, 2 * e, dtype=torch.bfloat16, device=dev)
    for i in range(2):
        dirs.append(dict(u=rnd(b, l, e).bfloat16(), A=-torch.exp(rnd(e, 16) * 0.3), D=rnd(e), delta_bias=rnd(e) - 1,
                         dt_weight=ops.pad_dt_weight(rnd(e, rank) * 0.3), xdbl=(rnd(b, l, P + 32) * 0.5).bfloat16(),
                         out=ycat[:, :, i * e:(i + 1) * e], reverse=bool(i)))
    auto = _native.lib().cm_scan_cl_fwd_auto_chunks(b, l, e, 2)
    alg = (4 * e + 2 * 16) * 2 * b * l * 2          # (4E + 2N) * s bytes per step per direction
    for c in chunk_list:
        for _ in range(3):
            ops.scan_cl_fwd([dict(d) for d in dirs], z=z, time_chunks=c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.scan_cl_fwd([dict(d) for d in dirs], z=z, time_chunks=c)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"batch {b} x {l} steps x {e} ch, dt_rank {rank}: chunks {c:2d}{' (auto)' if c == auto else '       '} "
              f"{us:7.1f} us  {alg / us / 1e6:6.2f} TB/s algorithmic = {alg / us / 1e6 / 8:.3f} of HBM peak", flush=True)


if __name__ == "__main__":
    case(16, 1000, 512, 16, [1, 2, 4, 8])
    case(8, 1000, 512, 16, [1, 4, 8, 16])
    case(4, 4000, 1024, 32, [1, 4, 8, 16])
    case(1, 4000, 512, 16, [1, 8, 16, 31])
    case(32, 1000, 512, 16, [1, 2, 4])
    case(64, 1000, 512, 16, [1, 2])
