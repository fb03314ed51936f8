#!/usr/bin/env python3
"""Headline benchmark: ConMamba-large CTC encoder forward, audio-frames/sec (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic batch that is already resident in HBM:
wav (B x 40 s @ 16 kHz) -> Fbank -> global normalisation -> CNN front end -> src Linear -> 18 ConMamba
layers -> final LayerNorm (reference train_CTC.py:285-298 + TransformerASR.encode), bf16 autocast, eval.
Each rank processes its own utterance shard (weak scaling, no data-path collective: SURVEY.md §8e).
Rank 0 prints ONE JSON line; `roofline` is for the dominant kernel (the selective scan) measured with HIP
events on the launching stream; `cpu_baseline` is the CPU oracle (port of the reference path) on a bounded
sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="conmamba_large_ctc")
    ap.add_argument("--batch", type=int, default=64,
                    help="utterances per GPU (40 s each).  The reference recipe fills 45-80 GB GPUs with 850-1700 s of audio "
                         "per batch (hparams/CTC/conmamba_large.yaml:108-113); 64 x 40 s = 2560 s is the same fill of a "
                         "288 GB MI355X.  SURVEY §8d's 16 x 40 s: --batch 16")
    ap.add_argument("--frames", type=int, default=4000, help="10 ms audio frames per utterance (L)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--streams", type=int, default=None, help="parts / HIP streams of the fused encoder (default: CM_STREAMS or 4)")
    ap.add_argument("--stream-mode", choices=["join", "free"], default=None,
                    help="join (default): the scan runs once per layer on the whole batch; free: fully independent parts")
    ap.add_argument("--cpu-frames", type=int, default=4000, help="frames per utterance of the CPU-baseline sample")
    ap.add_argument("--cpu-batch", type=int, default=64, help="utterances in the CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline(model, cfg, frames, batch):
    """The oracle (CPU port of the reference path, oracle/conmamba_oracle.py) on `batch` utterances of `frames`
    audio frames (default: the same batch one GPU step processes): full Fbank -> CNN -> encoder forward, fp32,
    all host threads.  Checker code, timed here
    only as the reported baseline."""
    from oracle import conmamba_oracle as O
    from mamba_asr_amd.asr import synthetic_wavs, samples_for_frames
    try:
        O.load_c_oracle()
        scan = O.selective_scan_c
        scan_kind = "C scan (oracle/scan_oracle.c, OpenMP)"
    except OSError:
        scan = O.selective_scan
        scan_kind = "torch time loop"
    p = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    wav, lens = synthetic_wavs(batch, samples_for_frames(frames), cfg.seed, "cpu")
    mean, std = p["normalize.glob_mean"], p["normalize.glob_std"]
    O.set_threads(O.host_threads(16))
    t0 = time.perf_counter()
    with torch.no_grad():
        out = O.asr_encode(p, wav, lens, cfg.num_encoder_layers, mean, std, scan=scan, n_fft=cfg.n_fft,
                           win_ms=cfg.win_length)
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    return {"value": round(batch * frames / dt, 1), "unit": "audio-frames/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{batch} utterances x {frames} frames ({frames / 100:.0f} s audio each), one full "
                      f"frontend+encoder forward, fp32, {scan_kind}, {dt:.1f} s wall"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)     # nccl == RCCL on ROCm
    from mamba_asr_amd import ops
    from mamba_asr_amd.asr import CONFIGS, ConMambaASR, samples_for_frames, synthetic_wavs

    from mamba_asr_amd import fused
    if a.streams is not None:
        fused.N_STREAMS = a.streams
    if a.stream_mode is not None:
        fused.STREAM_MODE = a.stream_mode
    cfg = CONFIGS[a.config]
    model = ConMambaASR(cfg).to(dev).eval()
    wavs, lens = synthetic_wavs(a.batch, samples_for_frames(a.frames), cfg.seed + rank, dev)
    amp = torch.bfloat16 if a.dtype == "bf16" else None

    def eager_step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
            return model.encode(wavs, lens)

    eager_step()                                            # first batch: fills the global normalisation statistics
    if a.no_graph:
        step = eager_step
    else:
        from mamba_asr_amd.fused import GraphedEncode
        graphed = GraphedEncode(model, wavs, lens, dtype=torch.bfloat16 if amp is not None else torch.float32)
        step = lambda: graphed()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(a.warmup, 1)):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)            # MAX over ranks
        elapsed = float(t.item())
    assert torch.isfinite(out.float()).all()
    frames_total = world * a.batch * a.frames * a.steps
    value = frames_total / elapsed

    # ---- roofline of the dominant kernel: selective-scan forward, HIP events on its launch stream
    roof = None
    if rank == 0:
        ops.LAUNCH_LOG = []
        for _ in range(3):
            eager_step()                                    # eager pass: each native launch bracketed by HIP events
        torch.cuda.synchronize()
        log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
        scans = [(e0.elapsed_time(e1), units) for name, e0, e1, units in log if name == "cm_scan_cl_fwd"]
        if scans:
            e_inner, n_state, s = cfg.expand * cfg.d_model, cfg.d_state, (2 if amp is not None else 4)
            avg_ms = sum(t for t, _ in scans) / len(scans)
            units = scans[0][1]                                      # scan steps (batch * T * directions) per launch
            alg_bytes = units * (4 * e_inner + 2 * n_state) * s      # (4E+2N)*s per scan step per direction
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            # HBM traffic of one launch from the PMC passes (FETCH_SIZE, WRITE_SIZE collected separately with rocprofv3
            # --pmc on tools/pmc_scan.py, profiles/): valid for the configuration it was measured on only
            traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", "r01", "pmc_scan_rows.json")) as f:
                    pmc = json.load(f)
                for run in pmc["runs"]:
                    w = run["workload"]
                    if (w["batch"], w["seqlen"], w["dim"], w["dtype"]) == (units // (2 * (a.frames // 4)), a.frames // 4, e_inner,
                                                                              "bf16" if amp is not None else "f32"):
                        traffic = run["traffic_bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                pass
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": "scan_rows_fwd_kernel (cm_scan_cl_fwd, xdbl mode: both BiMamba directions per launch)", "avg_launch_us": round(avg_ms * 1e3, 1),
                    "launches_per_step": len(scans) // 3, "alg_bytes_per_launch": alg_bytes}

    base = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        base = cpu_baseline(model, cfg, a.cpu_frames, a.cpu_batch)

    if rank == 0:
        # whole-path roofline position (SURVEY §8d canonical bytes per audio frame, bf16)
        bytes_per_frame = {"conmamba_large_ctc": 168464, "conmamba_small_ctc": 74568,
                           "conmambamamba_large_s2s": 226564}.get(a.config)
        line = {
            "metric": ("encoder audio-frames/sec (ConMamba-large, L=4000)" if (a.config, a.frames) == ("conmamba_large_ctc", 4000)
                       else f"encoder audio-frames/sec ({a.config}, L={a.frames})"), "value": round(value, 1),
            "unit": "audio-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if amp is not None else "f32", "data": "synthetic",
            "config": {"workload": f"{a.config}: encoder forward, {a.batch} utterances x {a.frames} frames "
                                   f"({a.frames // 4} scan steps) per GPU, random-init weights",
                       "global_batch": world * a.batch, "frames_per_utterance": a.frames,
                       "parallelism": f"utterance shards x{world} (no collective in forward)",
                       "launch": "eager" if a.no_graph else "hipGraph replay",
                       "streams": f"{min(fused.N_STREAMS, max(1, a.batch // 16))} ({fused.STREAM_MODE})"},
            "path_hbm_frac": None if bytes_per_frame is None else round(value / world * bytes_per_frame / 8e12, 4),
            "roofline": roof, "cpu_baseline": base,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
