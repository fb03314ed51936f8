#!/usr/bin/env python3
"""Headline benchmark: ConMamba-large CTC encoder forward, audio-frames/sec (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W           (N > 1: starts its own N ranks, one per GPU, as a child
                                                             `python -m torch.distributed.run`, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --mode train [--gpus N]                 data-parallel TRAINING step (fwd + bwd + AdamW, grad-accum 4,
                                                             gradient all-reduce over RCCL): secondary figure

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
import socket
import subprocess
import sys
import time

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
    ap.add_argument("--no-extras", action="store_true",
                    help="forward mode, 1 GPU: skip the extra keys of the line (\"train\": the training micro-batch of --mode train at 32 utterances; "
                         "\"survey_b16\": the forward at SURVEY §8d's 16 x 40 s)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--graph-train", action="store_true",
                    help="--mode train: replay forward + loss + backward of a micro-batch from a hipGraph (brain.Brain graph_steps)")
    ap.add_argument("--streams", type=int, default=None, help="parts / HIP streams of the fused encoder (default: CM_STREAMS or 2)")
    ap.add_argument("--stream-mode", choices=["join", "free", "pair"], default=None,
                    help="join (default): the scan runs once per layer on the whole batch; free: fully independent parts")
    ap.add_argument("--cpu-frames", type=int, default=4000, help="frames per utterance of the CPU-baseline sample")
    ap.add_argument("--cpu-batch", type=int, default=64, help="utterances in the CPU-baseline sample")
    ap.add_argument("--mode", choices=["forward", "train"], default="forward",
                    help="forward (default, the headline metric) or train: fwd + bwd + AdamW through the Brain loop with the "
                         "gradient exchange of mamba_asr_amd.ddp (reference train_CTC.py:1062 + conmamba_large.yaml:90)")
    ap.add_argument("--lens", choices=["full", "uniform"], default="full",
                    help="utterance lengths: full (wav_lens = 1, the roofline run) or uniform (relative lengths ~U(0.5, 1), "
                         "zero padded to --frames: SURVEY §8d's ragged run)")
    ap.add_argument("--accum", type=int, default=4, help="train mode: grad_accumulation_factor (conmamba_large.yaml:90)")
    ap.add_argument("--comm-dtype", choices=["f32", "bf16"], default="f32", help="train mode: gradient transport dtype")
    ap.add_argument("--ddp-algo", choices=["allreduce", "mesh"], default=None, help="train mode: see mamba_asr_amd/ddp.py")
    return ap.parse_args()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) as a CHILD
    `python -m torch.distributed.run` and return its exit code.  Runs before anything touches the GPU in this process
    (a process that has initialised the GPU must not be replaced; a child process is fine)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


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


def pmc_traffic(batch, seqlen, dim, dtype):
    """HBM bytes of one scan launch from the PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc
    runs of tools/pmc_scan.py, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes; summaries committed under
    profiles/rNN/): newest round first; valid only for the configuration it was measured on, else None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_scan_rows.json")), reverse=True):
        try:
            with open(path) as f:
                for run in json.load(f)["runs"]:
                    w = run["workload"]
                    if (w["batch"], w["seqlen"], w["dim"], w["dtype"]) == (batch, seqlen, dim, dtype):
                        return run["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def make_batch(a, cfg, rank, dev):
    """Synthetic 16 kHz utterances resident in HBM: full length, or relative lengths ~U(0.5, 1) with the
    tail zero padded (what the reference's padded batches look like; ConMamba applies no padding mask, Conmamba.py:635)."""
    from mamba_asr_amd.asr import samples_for_frames, synthetic_wavs
    wavs, lens = synthetic_wavs(a.batch, samples_for_frames(a.frames), cfg.seed + rank, dev)
    if a.lens == "uniform":
        g = torch.Generator().manual_seed(cfg.seed + 1000 + rank)
        rel = 0.5 + 0.5 * torch.rand(a.batch, generator=g)
        rel[0] = 1.0                                           # a padded batch always holds one full-length utterance
        n = wavs.shape[1]
        keep = torch.arange(n)[None, :] < torch.round(rel * n)[:, None]
        wavs = wavs * keep.to(dev)
        lens = rel.to(dev)
    return wavs, lens


def run_train(a, cfg, dev, rank, world, use_dist, emit_line=True):
    """Data-parallel training step (BASELINE.json config 4; reference train_CTC.py fit_batch through speechbrain's Brain +
    DDP, hparams/CTC/conmamba_large.yaml:84-91, 243-252): bf16 autocast, SpecAugment, CTC loss, backward, and every
    ``--accum``-th micro-batch the gradient exchange (mamba_asr_amd.ddp over RCCL), clip 5.0, AdamW, Noam.
    A step here = one micro-batch; `value` = audio frames all ranks push through fwd + bwd per second."""
    import torch.distributed as dist
    from mamba_asr_amd import ops
    from mamba_asr_amd import sb_compat as sb
    from mamba_asr_amd.asr import ConMambaASR
    from mamba_asr_amd.brain import Brain, Stage
    from mamba_asr_amd.ddp import GradAllReducer
    model = ConMambaASR(cfg).to(dev)
    aug = sb.Augmenter(augmentations=[sb.SpectrogramDrop(6, 12, 1, 5, "mean", 1), sb.SpectrogramDrop(10, 20, 1, 3, "mean", 2)])

    s2s = cfg.num_decoder_layers > 0                          # config 5: Mamba decoder + joint CTC / label-smoothed KL loss

    class ASR(Brain):
        def graph_prologue(self, batch):
            # --graph-train: Fbank, the running normalisation statistics and SpecAugment (host-side random draws) stay eager;
            # the captured region starts at the CNN front end with the augmented features as its input
            wavs, lens, tokens, tlens = batch
            with torch.no_grad():
                feats = self.modules["asr"].features(wavs, lens, epoch=0, augment=aug)
            return (feats, lens, tokens, tlens)

        def compute_forward(self, batch, stage):
            wavs, lens, tokens, tlens = batch
            m, a_ = self.modules["asr"], (aug if stage == Stage.TRAIN else None)
            feats = wavs if wavs.dim() == 3 else None          # (batch, frames, mels): graph_prologue's output
            if s2s:                                            # train_S2S.py:285-320: <bos> + tokens into the decoder
                bos = torch.cat([torch.ones_like(tokens[:, :1]), tokens], 1)
                return m.forward_s2s(wavs, lens, bos, epoch=0, augment=a_, feats=feats)
            return m.forward_ctc(wavs, lens, epoch=0, augment=a_, feats=feats)

        def compute_objectives(self, pred, batch, stage):
            wavs, lens, tokens, tlens = batch
            m = self.modules["asr"]
            if s2s:                                            # train_S2S.py:518-529: 0.3 CTC + 0.7 KL against tokens + <eos>
                p_ctc, p_seq = pred
                eos = torch.cat([tokens, torch.full_like(tokens[:, :1], 2)], 1)
                return m.s2s_objective(p_ctc.float(), p_seq.float(), tokens, tlens, eos, tlens, lens)
            return m.ctc_objective(pred, tokens, lens, tlens)

        def on_fit_batch_end(self, batch, outputs, loss, should_step):
            if should_step:
                self.hparams.noam(self.optimizer)

    brain = ASR({"asr": model}, opt_class=lambda ps: torch.optim.AdamW(ps, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=5e-4),
                hparams={"precision": "bf16" if a.dtype == "bf16" else "fp32", "grad_accumulation_factor": a.accum,
                         "max_grad_norm": 5.0, "noam": sb.NoamScheduler(1e-3, 7500)},
                run_opts={"device": str(dev), "graph_steps": bool(a.graph_train)})
    params = [p for p in brain.modules.parameters() if p.requires_grad]
    if use_dist:                                               # also at world 1: the exchange then runs through RCCL alone
        brain.reducer = GradAllReducer(params, comm_dtype=torch.bfloat16 if a.comm_dtype == "bf16" else None, algo=a.ddp_algo,
                                       always_exchange=True)
    brain.on_fit_start()
    brain.modules.train()
    wavs, lens = make_batch(a, cfg, rank, dev)
    g = torch.Generator().manual_seed(rank)
    # CTC recipe: characters, ~12.5 per second; S2S recipe: 5000 word pieces, ~2.5 per second (S ~ 400 at 160 s: SURVEY §8d)
    n_tok = a.frames // 40 if s2s else a.frames // 8
    tokens = torch.randint(3, cfg.output_neurons, (a.batch, n_tok), generator=g).to(dev)
    batch = (wavs, lens, tokens, lens.clone())

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # graph mode: the first micro-batch is eager, the second captures the "warm" variant, the first one after an optimizer step the
    # "fresh" one: both captures belong to warm-up
    n_warm = max(a.warmup, a.accum + 2 if a.graph_train else 1)
    losses = [brain.fit_batch(batch) for _ in range(n_warm)]
    fence()
    t0 = time.perf_counter()
    exposed = []
    for _ in range(a.steps):
        losses.append(brain.fit_batch(batch))
        if brain.reducer is not None and brain.step % a.accum == 0:
            exposed.append(brain.reducer.exposed_s)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    losses = [float(l) for l in losses]
    assert all(l == l for l in losses), "non-finite loss"
    value = world * a.batch * a.frames * a.steps / elapsed
    roof = None
    # backward-scan roofline: one more micro-batch on EVERY rank (it may end in the exchange), HIP events on rank 0
    ops.LAUNCH_LOG = [] if rank == 0 else None
    brain.graph_steps = False                                  # the per-launch event pairs need eager launches
    brain.fit_batch(batch)
    torch.cuda.synchronize()
    log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    if rank == 0:
        e_inner, n_state, s_ = cfg.expand * cfg.d_model, cfg.d_state, (2 if a.dtype == "bf16" else 4)
        per_step = (9 * e_inner + 4 * n_state) * s_                                 # fwd reads + dout, du / ddelta / dz + dB / dC, per scan step and direction
        for name, label, ndir in (("cm_scan_cl_bwd", "scan_rows_bwd_kernel + fixed-order reduce (cm_scan_cl_bwd: both BiMamba directions per launch)", 2),
                                  ("cm_selective_scan_bwd", "scan_bwd_kernel (cm_selective_scan_bwd, one direction per launch)", 1)):
            # encoder launches only (a decoder's scans have other sizes): the most frequent unit count
            ts_all = [(e0.elapsed_time(e1), u) for nm, e0, e1, u in log if nm == name]
            if not ts_all:
                continue
            u_enc = a.batch * (a.frames // 4) * ndir
            ts = [t for t, u in ts_all if u == u_enc] or [t for t, _ in ts_all]
            alg = u_enc * per_step
            avg_ms = sum(ts) / len(ts)
            roof = {"bound": "hbm", "achieved": round(alg / (avg_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "kernel": label,
                    "avg_launch_us": round(avg_ms * 1e3, 1), "us_per_direction": round(avg_ms * 1e3 / ndir, 1),
                    "launches_per_step": len(ts), "alg_bytes_per_launch": alg}
            break
    if rank == 0:
        line = {"metric": f"training audio-frames/sec ({cfg.name}, L={a.frames}, fwd+bwd+AdamW" + (", encoder + Mamba decoder, 0.3 CTC + 0.7 KL)" if s2s else ")"),
                "value": round(value, 1),
                "unit": "audio-frames/s", "n_gpus": world, "steps": a.steps, "warmup": n_warm,
                "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": a.dtype, "data": "synthetic",
                "config": {"workload": f"{cfg.name}: {'S2S' if s2s else 'CTC'} training micro-batch, {a.batch} utterances x {a.frames} frames per GPU, "
                                       f"SpecAugment, grad_accumulation_factor {a.accum}, AdamW + Noam, clip 5.0",
                           "global_batch": world * a.batch, "frames_per_utterance": a.frames,
                           "launch": "hipGraph replay of forward + loss + backward per micro-batch (exchange, clip, AdamW eager)" if a.graph_train else "eager",
                           "parallelism": f"dp{world}: gradient exchange every {a.accum} micro-batches "
                                          f"({'RCCL ' + (brain.reducer.algo if brain.reducer else '') if use_dist else 'none: single process'})"},
                "allreduce_bytes_per_optimizer_step": brain.reducer.bytes_per_step() if brain.reducer else 0,
                "exposed_comm_ms_per_optimizer_step": round(1e3 * sum(exposed) / len(exposed), 3) if exposed else None,
                "optimizer_steps": brain.optimizer_step, "loss_first_last": [round(losses[0], 3), round(losses[-1], 3)],
                "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "roofline": roof, "cpu_baseline": None}
        if emit_line:
            emit(line)
        return line
    return None


_result_fd = None


def emit(line: dict) -> None:
    """The result line goes to the process's original stdout; everything else this process (or a library it loads: RCCL's
    version banner, libdrm's complaints) writes to file descriptor 1 has been pointed at stderr, so stdout carries
    exactly one line."""
    data = (json.dumps(line) + "\n").encode()
    if _result_fd is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_result_fd, data)


def main():
    a = parse()
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # started by torch.distributed.run
    if not launched and a.gpus > 1:
        sys.exit(launch_ranks(a))
    global _result_fd
    sys.stdout.flush()
    _result_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE is {world}: launch with --nproc-per-node {a.gpus} "
                         f"(or run `python bench.py --gpus {a.gpus}` without a launcher and it starts the ranks itself)")
    global torch
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = launched
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
    if a.mode == "train":
        run_train(a, cfg, dev, rank, world, use_dist)
        if use_dist:
            dist.destroy_process_group()
        return
    model = ConMambaASR(cfg).to(dev).eval()
    wavs, lens = make_batch(a, cfg, rank, dev)
    amp = torch.bfloat16 if a.dtype == "bf16" else None

    def eager_step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
            return model.encode(wavs, lens)

    model.calibrate(wavs, lens)                             # first batch: fills the global normalisation statistics
    eager_step()
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
        # Eager pass, each native launch bracketed by HIP events on its stream.  An event's timestamp is taken when the stream
        # REACHES it: if the host is the slower side, the stream sits empty, the first event fires at once and the interval
        # includes the host's launch latency (measured: +11 us on a 250 us kernel).  So every pass starts behind a blocker --
        # device-to-device copies long enough for the host to enqueue the whole step -- and the kernels then run back to back
        # as they do in the timed hipGraph replay.
        blk = torch.empty(2, 1 << 28, dtype=torch.float32, device=dev)                       # 2 x 1 GiB
        t0h = time.perf_counter()
        eager_step()
        torch.cuda.synchronize()
        host_s = time.perf_counter() - t0h                                                     # upper bound of the host's enqueue time
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        blk[0].copy_(blk[1])
        c1.record()
        torch.cuda.synchronize()
        ncopy = max(2, int(1.5 * host_s / max(c0.elapsed_time(c1) * 1e-3, 1e-5)) + 1)
        # An event pair around a launch also spans the dispatch of a dependent kernel and the end-of-kernel signal, which
        # rocprofv3's kernel timestamps do not.  An EMPTY pair (two records back to back, queued behind the same step) measures
        # what the pair itself spans; that is reported and taken off the scan's interval (round 2 put a one-element kernel
        # between the records and so also subtracted that kernel's ~2 us of run time: frac 0.261 against rocprofv3's 0.257).
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3 * 8)]
        ops.LAUNCH_LOG = []
        for rep in range(3):
            for _ in range(min(ncopy, 64)):
                blk[0].copy_(blk[1])
            eager_step()
            for q0, q1 in pairs[8 * rep:8 * rep + 8]:
                q0.record()                                   # nothing between the two records: the interval an event pair
                q1.record()                                   # itself spans in this queue (not a kernel's run time)
        torch.cuda.synchronize()
        log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
        pair_ms = sorted(q0.elapsed_time(q1) for q0, q1 in pairs)
        pair_overhead_ms = pair_ms[len(pair_ms) // 2]                                          # median of the empty intervals
        del blk
        scans = [(e0.elapsed_time(e1), units) for name, e0, e1, units in log if name == "cm_scan_cl_fwd"]
        if scans:
            e_inner, n_state, s = cfg.expand * cfg.d_model, cfg.d_state, (2 if amp is not None else 4)
            raw_ms = sum(t for t, _ in scans) / len(scans)
            avg_ms = raw_ms - pair_overhead_ms
            units = scans[0][1]                                      # scan steps (batch * T * directions) per launch
            alg_bytes = units * (4 * e_inner + 2 * n_state) * s      # (4E+2N)*s per scan step per direction
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            # HBM traffic of one launch from the PMC passes (FETCH_SIZE, WRITE_SIZE collected separately with rocprofv3
            # --pmc on tools/pmc_scan.py, profiles/): valid for the configuration it was measured on only
            traffic = pmc_traffic(units // (2 * (a.frames // 4)), a.frames // 4, e_inner, "bf16" if amp is not None else "f32")
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": "scan_rows_fwd_kernel (cm_scan_cl_fwd, xdbl mode: both BiMamba directions per launch)", "avg_launch_us": round(avg_ms * 1e3, 1),
                    "event_interval_us": round(raw_ms * 1e3, 1), "event_pair_overhead_us": round(pair_overhead_ms * 1e3, 1),
                    "launches_per_step": len(scans) // 3, "alg_bytes_per_launch": alg_bytes}

    # ---- extra keys (1 GPU, the default configuration): SURVEY §8d's 16 x 40 s forward, and the training micro-batch
    extras = {}
    if rank == 0 and world == 1 and not a.no_extras and a.config == "conmamba_large_ctc" and a.batch != 16 and not a.no_graph:
        import copy
        a16 = copy.copy(a)
        a16.batch = 16
        w16, l16 = make_batch(a16, cfg, rank, dev)
        g16 = GraphedEncode(model, w16, l16, dtype=torch.bfloat16 if amp is not None else torch.float32)
        for _ in range(3):
            g16()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            g16()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ops.LAUNCH_LOG = []
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
            for _ in range(2):
                model.encode(w16, l16)
        torch.cuda.synchronize()
        log16, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
        sc = [e0.elapsed_time(e1) for nm, e0, e1, _ in log16 if nm == "cm_scan_cl_fwd"]
        sc = sc[len(sc) // 2:]                                                       # the second pass
        s_ = 2 if amp is not None else 4
        alg16 = 16 * (a.frames // 4) * 2 * (4 * cfg.expand * cfg.d_model + 2 * cfg.d_state) * s_
        extras["survey_b16"] = {"workload": f"16 utterances x {a.frames} frames (SURVEY §8d's batch)", "value": round(16 * a.frames * a.steps / el, 1),
                                "ms_per_step": round(el / a.steps * 1e3, 3),
                                "scan_event_interval_us": round(sum(sc) / len(sc) * 1e3, 1) if sc else None,
                                "scan_roofline_frac_raw_interval": round(alg16 / (sum(sc) / len(sc) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sc else None}
        del g16, w16, l16
    base = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        base = cpu_baseline(model, cfg, a.cpu_frames, a.cpu_batch)
    if rank == 0 and world == 1 and not a.no_extras and a.config == "conmamba_large_ctc":
        import copy
        at = copy.copy(a)
        at.batch, at.steps, at.warmup, at.accum, at.comm_dtype, at.ddp_algo = 32, 8, 4, 4, "f32", None
        at.graph_train = not a.no_graph                                              # the micro-batch's fwd + loss + bwd as a hipGraph, as the forward line
        graphed = step = out = None                                                  # release the forward's graph and buffers
        del model
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        try:                                                                         # an extra key must not take the headline line down with it
            tl = run_train(at, cfg, dev, rank, world, False, emit_line=False)
            extras["train"] = {k: tl[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "peak_mem_gib", "optimizer_steps", "roofline")}
            extras["train"]["workload"] = tl["config"]["workload"]
            extras["train"]["launch"] = tl["config"]["launch"]
        except Exception as exc:                                                     # noqa: BLE001 -- reported in the line, never silent
            extras["train"] = {"error": f"{type(exc).__name__}: {exc}"[:500]}

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
                                   f"({a.frames // 4} scan steps) per GPU, random-init weights"
                                   + ("" if a.lens == "full" else ", relative lengths ~U(0.5, 1) zero padded"),
                       "global_batch": world * a.batch, "frames_per_utterance": a.frames,
                       "parallelism": f"utterance shards x{world} (no collective in forward)",
                       "launch": "eager" if a.no_graph else "hipGraph replay",
                       "streams": f"{min(fused.N_STREAMS, max(1, a.batch // 16))} ({fused.STREAM_MODE})"},
            "path_hbm_frac": None if bytes_per_frame is None else round(value / world * bytes_per_frame / 8e12, 4),
            "roofline": roof, "cpu_baseline": base,
        }
        line.update(extras)
        emit(line)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
