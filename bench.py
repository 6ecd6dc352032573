#!/usr/bin/env python3
"""Headline benchmark: ACT policy steps/sec on synthetic 4-camera 480x640 batches (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--mode infer|train|eval-shard]

N > 1: either launched by  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...  (WORLD_SIZE set:
this process is one rank), or started plainly as  python bench.py --gpus N  -- then this process is only a LAUNCHER: it
starts N fresh rank processes (one per GPU, rendezvous on 127.0.0.1) before touching the GPU itself, relays rank 0's JSON
line and exits non-zero if any rank failed (or if the box has fewer than N GPUs).

A "step" is one batched policy query through the C-ABI HIP path: u8 NHWC images already resident in HBM ->
multi-camera ResNet18 -> DETR encoder/decoder -> a_hat [B,100,16] -> temporal-ensemble reduction.  Weights are
random-init of the reference architecture (no network for checkpoints), inputs synthetic; work per step is
independent of the values.  N > 1 runs N independent replicas on disjoint batches (episodes shard embarrassingly;
the path has no data-path collective) and reports the whole-job aggregate with max-over-ranks timing.

Rank 0 prints ONE JSON line with the driver's contract plus:
  roofline     dominant kernel (the instantiation with the largest total time): achieved = algorithmic FLOPs of its
               launches / their summed duration, measured with HIP events on the launch stream; peak = the dense MFMA
               peak of the arithmetic the kernel uses (see kernel_peak); traffic = fabric bytes per launch from the PMC
               passes committed under profiles/ (traffic_source names the file: it is NOT produced by this run)
  sustained    a second timed leg of >= 10 s of back-to-back steps (DVFS: sustained MFMA load clocks lower than a burst)
  with_h2d     the same step fed from pinned host memory each step (fresh frames cross PCIe: never the headline value)
  extra        driver-visible sub-records of the other BASELINE configurations on one GPU: b1 (the reference's own
               rollout mode, imitate_episodes.py:397), b50 (config 2's batch), native_fp32 (ACTMI_GEMM_PREC=f32: every
               product on the exact fp32 MFMA), train_b64 (config 3: forward+backward+AdamW at batch 64)
  cpu_baseline the CPU oracle (torch fp32 restatement of the reference, as written: all 7 decoder layers) timed on
               this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).

--mode eval-shard is BASELINE config 5: sim_insertion_scripted poses, 50 episodes per GPU stepped in lock-step on the
SyntheticEnv stand-in, ONE all-gather (RCCL) of (episode_return, highest_reward); value = policy steps/s of the whole job.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "act-plus-plus_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_MATRIX_TFLOPS = 157.3          # MI355X_MICROARCH.md, chip-level parameters
PEAK_FP16_MATRIX_TFLOPS = 16 * 157.3     # dense F16/BF16 MFMA = 16x the f32 MFMA rate (same guide, matrix-core table)
# PMC traffic files, newest first: the first one that exists is read; its own stamp (commit + hash of the kernel sources it
# was measured on) is printed next to the number and compared with the live tree (roofline.traffic_stale)
TRAFFIC_FILES = [os.path.join("profiles", f) for f in ("r03_traffic.json", "r02_traffic.json", "r01_j_traffic.json")]

# "dtype" of the JSON line: the type the path stores and accumulates in, with the way products are formed spelled out
DTYPE = {"f16x3": "f32 (f16x3 split products: 3 fp16 MFMAs per fp32 product, fp32 accumulate)", "f32": "f32"}

ARITH = {
    "f16x3": "fp32 in / fp32 out / fp32 accumulate; every GEMM, convolution (incl. the 7x7 stem) and attention product is "
             "formed on the fp16 matrix pipe from operands split exactly into two fp16 pieces (3 MFMA products per fp32 "
             "product, fp32-grade parity); LayerNorm / softmax / pooling fp32 VALU",
    "f32": "fp32 everywhere, native fp32 MFMA (v_mfma_f32_32x32x2_f32)",
}
GFLOP_PER_SAMPLE_LIVE = 145.4            # work that influences the output (decoder layer 0 only), C=4
GFLOP_PER_SAMPLE_AS_WRITTEN = 160.4
GFLOP_TRAIN_PER_SAMPLE_LIVE = 447.0      # fwd 149 incl. CVAE + bwd ~2x


def kernel_peak(name):
    """Dense MFMA peak, in ALGORITHMIC fp32 FLOP/s, of the arithmetic a kernel uses.  *_f16x3 kernels form every fp32
    product from three fp16 MFMA products (exact two-piece fp16 split of both operands), so their ceiling is a third of
    the fp16 matrix peak; everything else runs the native fp32 MFMA."""
    if "f16x3" in name:
        return PEAK_FP16_MATRIX_TFLOPS / 3.0, "f16 MFMA dense peak / 3 products per fp32 product"
    if "bf16" in name:
        return PEAK_FP16_MATRIX_TFLOPS, "bf16 MFMA dense peak (one product per fp32 product)"
    return PEAK_FP32_MATRIX_TFLOPS, "f32 MFMA dense peak"


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# --------------------------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N with no WORLD_SIZE -> N child ranks.  Runs BEFORE anything initialises the GPU
# in this process (device_count() does not on this image) and never re-execs.
# --------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    import socket
    import torch
    ndev = torch.cuda.device_count()
    if ndev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this box has {ndev} GPU(s); nothing was run", file=sys.stderr)
        return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    import threading
    procs = []
    # rank 0's stdout goes to a file (a pipe nobody drains could block it); the others' is dropped
    out_f = tempfile.TemporaryFile(mode="w+")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out_f if r == 0 else subprocess.DEVNULL, text=True))
    # supervise ALL ranks: the first non-zero exit (or the overall limit) ends the others -- a rank that died at start would
    # otherwise leave rank 0 in init_process_group / a barrier until the RCCL timeout, holding the GPUs meanwhile
    deadline = time.time() + float(os.environ.get("ACTMI_BENCH_LAUNCH_TIMEOUT_S", "3000"))
    rcs = [None] * len(procs)
    failed = False
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs) or time.time() > deadline:
            failed = True
            break
        time.sleep(0.2)
    if failed:
        live = [p for p, rc in zip(procs, rcs) if rc is None]
        for p in live:
            p.terminate()                       # the exact children started above, never a pattern
        t_kill = time.time() + 10.0
        for p in live:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        rcs = [p.returncode for p in procs]
    out_f.seek(0)
    out0 = out_f.read()
    out_f.close()
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if failed or any(rcs) or line is None:
        print(f"bench.py: rank exit codes {rcs}{' (siblings of the first failure were terminated)' if failed else ''}; "
              f"rank 0 printed {'no' if line is None else 'a'} JSON line", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (metric is quoted at 8; train default 64)")
    ap.add_argument("--mode", choices=["infer", "train", "eval-shard"], default="infer",
                    help="infer = headline policy-query metric; train = ACT training step (forward+backward+AdamW); "
                         "eval-shard = episode-sharded eval rollouts with one RCCL all-gather (BASELINE config 5)")
    ap.add_argument("--graph", action="store_true", default=True,
                    help="(default) replay the step as ONE captured hipGraph; per-kernel events then come from extra eager "
                         "steps after the timed region, since events cannot bracket kernels inside a graph")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch the step's kernels eagerly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-extras", action="store_true", help="skip the b1 / b50 / native_fp32 / train_b64 sub-records")
    ap.add_argument("--sustained-s", type=float, default=10.0, help="length of the sustained leg in seconds (0 = skip)")
    ap.add_argument("--shapes", action="store_true",
                    help="per-shape GEMM table (M,N,K,groups,split,workgroups,us,TFLOP/s per distinct launch) in 'shapes'")
    ap.add_argument("--train-prec", choices=["f16x3", "bf16"], default=None,
                    help="--mode train: arithmetic of the step's GEMMs (default f16x3 = fp32-grade; bf16 = BASELINE config 3 as written)")
    ap.add_argument("--episodes-per-gpu", type=int, default=50, help="eval-shard: episodes per rank (config 5: 50)")
    ap.add_argument("--episode-len", type=int, default=None, help="eval-shard: timesteps per episode (task default 400)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if args.shapes:
        os.environ["ACTMI_PROF_SHAPES"] = "1"            # read once by the library at its first profiled launch

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} needs GPU {local_rank}; {torch.cuda.device_count()} visible", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)          # "nccl" is RCCL on ROCm

    from actmi.config import ACTConfig
    cfg = ACTConfig()                                           # C=4, 480x640, Q=100, D=512, F=3200, 4 enc + 7 dec
    ctx = {"rank": rank, "world": world, "dev": dev, "dist": dist}
    if args.mode == "train":
        out = bench_train(args, cfg, args.batch or 64, ctx)
    elif args.mode == "eval-shard":
        out = bench_eval_shard(args, ctx)
    else:
        out = bench_infer(args, cfg, args.batch or 8, ctx)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


# --------------------------------------------------------------------------------------------------------------
# inference
# --------------------------------------------------------------------------------------------------------------
class InferRig:
    """Engine + resident synthetic inputs + (optionally) the captured graph of one step at batch B."""

    def __init__(self, cfg, B, dev, seed, prec=None, graph=True):
        import torch
        from actmi import weights as W
        from actmi import ops
        from actmi.engine import ACTEngine
        self.torch, self.B, self.dev = torch, B, dev
        self.eng = ACTEngine(cfg, max_batch=B, device=str(dev), gemm_prec=prec)
        self.eng.load_state_dict(W.generate_state_dict(cfg, seed=0))
        self.eng.finalize()
        inp = W.generate_inputs(cfg, B, seed=seed)
        self.qpos = torch.from_numpy(inp["qpos"]).to(dev)
        self.image_host = torch.from_numpy(inp["image_u8"]).pin_memory()
        self.image = self.image_host.to(dev)                    # resident in HBM before any timed region
        self.a_hat = torch.empty((B, cfg.num_queries, cfg.action_dim), dtype=torch.float32, device=dev)
        self.ens = ops.TemporalEnsemble(B, cfg.num_queries, cfg.action_dim, 0.01, dev)
        self.replay, self.pipe = None, None
        if graph:
            try:
                # forward + ensemble as ONE graph launch.  The with_h2d leg uses engine.InferPipeline instead: two input buffers,
                # the step as a trunk graph and a transformer graph, the H2D copy of the NEXT step's frames beside the latter.
                self.replay = self.eng.capture_infer(B, with_ensemble=self.ens)
            except Exception as e:                               # capture unsupported -> eager launches
                log(f"hipGraph capture failed ({e}); falling back to eager launches")
                self.replay, self.pipe = None, None
        if self.replay is not None:
            # the graph's static input buffers ARE the resident inputs (a deployment writes its frames there), so a step is the
            # graph launch alone
            s_qpos, s_img, _ = self.replay.static
            s_qpos.copy_(self.qpos); s_img.copy_(self.image)
            self.qpos_host = self.qpos.cpu().pin_memory()
            self.qpos, self.image = s_qpos, s_img
            torch.cuda.synchronize(dev)

    @property
    def graphed(self):
        return self.replay is not None

    def step(self, image=None):
        if self.replay is not None:
            if image is not None:                              # fresh host frames: straight into the graph's input buffer
                self.image.copy_(image, non_blocking=True)
            return self.replay(self.qpos, self.image)[1]
        image = self.image if image is None else image.to(self.dev, non_blocking=True)
        self.eng.forward_infer(self.qpos, image, out=self.a_hat)
        return self.ens.step(self.a_hat)

    def eager_step(self):
        self.eng.forward_infer(self.qpos, self.image, out=self.a_hat)
        return self.ens.step(self.a_hat)

    def output(self):
        return self.replay.static[2] if self.replay is not None else self.a_hat

    def timed(self, steps, warmup, barrier=None, with_h2d=False):
        """seconds for exactly `steps` steps, bracketed by barrier + synchronize on both sides.  with_h2d: every step's frames
        come from pinned host memory -- double-buffered when the step is graphed (the copy of step t + 1 runs beside step t)."""
        torch = self.torch
        sync = barrier or (lambda: torch.cuda.synchronize(self.dev))
        for _ in range(warmup):
            self.step()
        sync()
        if with_h2d and self.replay is not None and self.pipe is None:
            try:                                             # built on first use: the headline leg never sees its graphs or streams
                from actmi.engine import InferPipeline
                self.pipe = InferPipeline(self.eng, self.B, with_ensemble=self.ens)
            except Exception as e:                           # noqa: BLE001
                log(f"InferPipeline unavailable ({e}); the with_h2d leg copies in front of every step")
        if with_h2d and self.pipe is not None:
            pipe, nxt = self.pipe, (self.qpos_host, self.image_host)
            t0 = time.perf_counter()
            pipe.feed(*nxt)
            for i in range(steps):
                pipe.step(next_inputs=nxt if i + 1 < steps else None)   # trunk graph, event, copy of the next frame, transformer graph
            sync()
            return time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(self.image_host if with_h2d else None)
        sync()
        return time.perf_counter() - t0

    def close(self):
        self.replay = self.pipe = None
        self.eng = None
        self.torch.cuda.empty_cache()


LAUNCH_NOTE = ("" if os.environ.get("ACTMI_CAM_PIPE") == "0" else
               "; the ResNet trunk runs as two concurrent branches of the graph (camera halves on two streams)")


def kernel_table(prof, steps):
    prof = [p for p in prof if p["ms"] > 0]
    prof.sort(key=lambda p: -p["ms"])
    gpu_ms = sum(p["ms"] for p in prof)
    rows = [{"name": p["name"], "launches_per_step": p["count"] / steps, "avg_us": p["ms"] * 1e3 / p["count"],
             "share": p["ms"] / gpu_ms, "tflops": (p["flops"] / (p["ms"] * 1e-3) / 1e12) if p["flops"] else None,
             "gbps": p["bytes"] / (p["ms"] * 1e-3) / 1e9} for p in prof]
    return prof, rows, gpu_ms


def dominant(prof):
    """(record, achieved TFLOP/s) of the kernel instantiation with the largest total time."""
    dom = prof[0]
    return dom, dom["flops"] / (dom["ms"] * 1e-3) / 1e12


def profile_eager(rig, steps):
    """per-kernel events over `steps` eager steps (events cannot bracket kernels inside a graph)."""
    from actmi import lib as L
    L.profile_enable(True)
    for _ in range(steps):
        rig.eager_step()
    rig.torch.cuda.synchronize(rig.dev)
    L.profile_enable(False)
    return L.profile_report()


def sub_record(cfg, B, dev, steps, warmup, prec=None, note="", env=None):
    """One driver-visible sub-record: ms/step, policy steps/s and the dominant kernel's roofline fraction at batch B.
    env: environment settings read by the engine at create (restored afterwards)."""
    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        rig = InferRig(cfg, B, dev, seed=4321 + B, prec=prec)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    dt = rig.timed(steps, warmup)
    prof, rows, gpu_ms = kernel_table(profile_eager(rig, max(3, steps // 2)), max(3, steps // 2))
    dom, ach = dominant(prof)
    lat = latency(rig, 30)
    assert rig.torch.isfinite(rig.output()).all()
    rig.eng.check_flags()
    graphed = rig.graphed
    rig.close()
    return {"per_gpu_batch": B, "steps": steps, "ms_per_step": dt / steps * 1e3, "policy_steps_per_s": B * steps / dt,
            "step_latency_ms": lat, "launch": "hipGraph replay" if graphed else "eager",
            "dominant_kernel": dom["name"], "dominant_tflops": ach, "dominant_frac": ach / kernel_peak(dom["name"])[0],
            "dominant_share": dom["ms"] / gpu_ms, "note": note}


def latency(rig, n):
    """p10 / p50 / p90 of single steps bracketed by events on the stream they run on."""
    torch = rig.torch
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in evs:
        e0.record()
        rig.step()
        e1.record()
    torch.cuda.synchronize(rig.dev)
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    return {"n": n, "p10": ts[int(0.1 * (n - 1))], "p50": ts[n // 2], "p90": ts[int(0.9 * (n - 1))]}


def bench_infer(args, cfg, B, ctx):
    import torch
    from actmi import lib as L
    rank, world, dev, dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["dist"]
    log(f"rank {rank}/{world}: building the engine (batch {B})")
    rig = InferRig(cfg, B, dev, seed=1234 + rank, graph=args.graph)
    log("engine ready")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        rig.step()
    barrier()
    log("warm-up done")
    L.profile_enable(not rig.graphed)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rig.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if rig.graphed:                    # per-kernel timing leg: the same number of steps, eager, outside the timed region
        prof = profile_eager(rig, args.steps)
    else:
        L.profile_enable(False)
        prof = L.profile_report()
    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(rig.output()).all()
    rig.eng.check_flags()              # default-on range guard of the f16x3 arithmetic (reads the device flag word)
    # which ranks took part: every rank's id through the eval path's own collective (all_gather_rows: RCCL when world > 1)
    from actmi import dist_utils
    ids = dist_utils.all_gather_rows(torch.full((1, 2), float(rank)), [1] * world)
    ranks_seen = sorted(int(x) for x in ids[:, 0].tolist())

    # sustained leg: >= args.sustained_s of back-to-back steps (every rank runs it; max over ranks)
    sustained = None
    if args.sustained_s > 0:
        n_sus = max(args.steps, int(args.sustained_s / (elapsed / args.steps)) + 1)
        dt = rig.timed(n_sus, 0, barrier)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        sustained = {"steps": n_sus, "seconds": dt, "ms_per_step": dt / n_sus * 1e3, "value": world * B * n_sus / dt,
                     "unit": "policy steps/s"}
        log(f"sustained leg: {n_sus} steps in {dt:.1f} s")
    if rank != 0:
        return None

    lat = latency(rig, max(10, min(200, args.steps)))
    h2d_dt = rig.timed(args.steps, 2, with_h2d=True)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    prof, kernels, gpu_ms = kernel_table(prof, args.steps)
    dom, ach = dominant(prof)
    peak, peak_is = kernel_peak(dom["name"])
    # fabric traffic per launch of the dominant kernel: measured in separate rocprofv3 --pmc passes (FETCH_SIZE x2 for the
    # gfx950 under-count, + WRITE_SIZE) of this same command and COMMITTED under profiles/ -- read from there, not produced here
    from actmi.buildinfo import kernel_source_sha16
    live_sha = kernel_source_sha16()
    traffic, traffic_source, traffic_commit, traffic_sha = None, None, None, None
    for f in TRAFFIC_FILES:
        try:
            tj = json.load(open(os.path.join(ROOT, f)))
            key = dom["name"].split("[")[0]
            traffic = tj["kernels"][key]["traffic_bytes_per_launch"] if B == 8 else None
            traffic_commit, traffic_sha = tj.get("commit"), tj.get("kernel_source_sha16")
            traffic_source = f + " (rocprofv3 --pmc passes of an earlier run of this command; not measured by this run)"
            break
        except Exception:
            continue
    # the two flavours of the same gemm.hip main loop (implicit-GEMM convolutions / row-major transformer products) carry 26 % of the
    # step each: which one is "dominant" flips between boxes, so the other one is reported beside it whenever it is within 10 %
    runner_up = None
    if len(prof) > 1 and prof[1]["ms"] >= 0.9 * dom["ms"] and prof[1]["flops"] > 0:
        r2 = prof[1]
        a2 = r2["flops"] / (r2["ms"] * 1e-3) / 1e12
        t2 = None
        try:
            t2 = tj["kernels"][r2["name"].split("[")[0]]["traffic_bytes_per_launch"] if (traffic is not None and B == 8) else None
        except Exception:                                    # noqa: BLE001
            pass
        runner_up = {"kernel": r2["name"], "achieved": a2, "frac": a2 / kernel_peak(r2["name"])[0], "share_of_dominant_time": r2["ms"] / dom["ms"],
                     "avg_launch_us": r2["ms"] * 1e3 / r2["count"], "launches_per_step": r2["count"] / args.steps,
                     "algorithmic_bytes_per_launch": r2["bytes"] / r2["count"], "traffic": t2}
    prec = os.environ.get("ACTMI_GEMM_PREC", "f16x3")
    out = {
        "metric": "policy steps/sec (4x480x640 cams, chunk=100, bs=8)",
        "value": value, "unit": "policy steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE["f32" if prec.startswith("f3") else "f16x3"], "data": "synthetic",
        "arithmetic": ARITH["f32" if prec.startswith("f3") else "f16x3"],
        "kernel_source_sha16": live_sha,
        "config": {"workload": "ACT eval policy query: 4 cams 480x640 u8, chunk 100, hidden 512, ff 3200, "
                               "4 enc + 7 dec layers (layer 0 live), per-GPU batch %d, + temporal ensemble" % B,
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"replicas x{world}",
                   "launch": ("hipGraph replay" if rig.graphed else "eager") + LAUNCH_NOTE},
        "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                     "measured": "per-launch HIP events on the launch stream, launches back to back on ONE stream (the profiled "
                                 "steps run the trunk as a single branch: events cannot bracket a launch that overlaps "
                                 "another branch's); extra.single_branch times the whole step that way",
                     "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_source, "traffic_commit": traffic_commit,
                     "traffic_kernel_source_sha16": traffic_sha,
                     # True: the kernels changed since the PMC passes were taken (or the file carries no stamp)
                     "traffic_stale": None if traffic is None else (traffic_sha != live_sha),
                     "peak_is": peak_is,
                     "frac_of_native_fp32_mfma_peak": ach / PEAK_FP32_MATRIX_TFLOPS,
                     "avg_launch_us": dom["ms"] * 1e3 / dom["count"], "launches_per_step": dom["count"] / args.steps,
                     "flop_per_launch": dom["flops"] / dom["count"],
                     "algorithmic_bytes_per_launch": dom["bytes"] / dom["count"], "runner_up": runner_up},
        "whole_step": {"gflop_per_sample_live": GFLOP_PER_SAMPLE_LIVE,
                       "achieved_tflops_live": value / world * GFLOP_PER_SAMPLE_LIVE / 1e3,
                       "frac_of_fp32_matrix_peak": value / world * GFLOP_PER_SAMPLE_LIVE / 1e3 / PEAK_FP32_MATRIX_TFLOPS,
                       "gpu_kernel_ms_per_step": gpu_ms / args.steps},
        "sustained": sustained,
        "with_h2d": {"ms_per_step": h2d_dt / args.steps * 1e3, "value": B * args.steps / h2d_dt, "unit": "policy steps/s",
                     "h2d_bytes_per_step": int(rig.image_host.numel()),
                     "note": "fresh u8 frames (and qpos) copied from pinned host memory every step, double-buffered: the copy "
                             "of step t+1 runs on a copy stream beside the transformer of step t, released by an event between the step's trunk graph and its transformer graph (engine.InferPipeline); "
                             "rank 0 only; not the headline"},
        "kernels": kernels,
        "step_latency_ms": lat,
        "ranks_seen": ranks_seen,
    }
    if args.shapes:
        out["shapes"] = shape_table(prof, args.steps)
    rig.close()
    if world == 1 and not args.no_extras:
        extra = {}
        for name, fn in (("b1", lambda: sub_record(cfg, 1, dev, 50, 10, note="the reference's rollout mode: one query per timestep")),
                         ("b50", lambda: sub_record(cfg, 50, dev, 6, 2, note="config 2's batch (50 parallel episodes), 4 cameras")),
                         ("native_fp32", lambda: sub_record(cfg, B, dev, 10, 2, prec="f32",
                                                            note="every product on the exact fp32 MFMA (gemm_prec=f32)")),
                         ("single_branch", lambda: sub_record(cfg, B, dev, args.steps, args.warmup, env={"ACTMI_CAM_PIPE": "0"},
                                                              note="ACTMI_CAM_PIPE=0: every launch of the ResNet trunk spans all "
                                                                   "cameras, one stream (the launch structure the roofline "
                                                                   "record and the rocprofv3 summaries describe)")),
                         ("train_b64", lambda: train_record(cfg, 64, dev, 4, 2)),
                         ("train_b64_bf16", lambda: train_record(cfg, 64, dev, 4, 2, train_prec="bf16")),
                         ("diffusion_b32", lambda: diffusion_record(dev, 32, 3, 1))):
            try:
                log(f"extra.{name}")
                extra[name] = fn()
            except Exception as e:           # a sub-record must not take the headline down; the failure is reported
                extra[name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        out["extra"] = extra
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, B, args.cpu_iters)
    return out


def shape_table(prof, steps):
    """--shapes: one row per distinct GEMM launch shape (the profiler's class names carry M,N,K,groups,split,workgroups)."""
    import re
    rows = []
    for p in prof:
        m = re.search(r"\[M=(\d+),N=(\d+),K=(\d+),g=(\d+),sk=(\d+),wgs=(\d+)\]", p["name"])
        if not m:
            continue
        M, N, K, g, sk, wgs = map(int, m.groups())
        rows.append({"kernel": p["name"].split("[")[0], "M": M, "N": N, "K": K, "groups": g, "splitk": sk, "workgroups": wgs,
                     "launches_per_step": p["count"] / steps, "avg_us": p["ms"] * 1e3 / p["count"],
                     "tflops": p["flops"] / (p["ms"] * 1e-3) / 1e12,
                     "frac": p["flops"] / (p["ms"] * 1e-3) / 1e12 / kernel_peak(p["name"])[0]})
    rows.sort(key=lambda r: -r["avg_us"] * r["launches_per_step"])
    return rows


# --------------------------------------------------------------------------------------------------------------
# training step (BASELINE config 3)
# --------------------------------------------------------------------------------------------------------------
def train_rig(cfg, B, dev, seed, world, train_prec=None):
    import torch
    from actmi import weights as W
    from actmi.engine import ACTEngine
    eng = ACTEngine(cfg, max_batch=B, device=str(dev), training=True, train_prec=train_prec)
    eng.load_state_dict(W.generate_state_dict(cfg, seed=0))
    eng.finalize()
    inp = W.generate_inputs(cfg, B, seed=seed, with_actions=True)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}

    def step(i, batch=None):
        eng.zero_grad()
        image, qpos, actions, is_pad = batch if batch is not None else (t["image_u8"], t["qpos"], t["actions"], t["is_pad"])
        out = eng.forward_train(qpos, image, actions, is_pad, eps=t["eps"])
        if world > 1 and os.environ.get("ACTMI_DP_MODE", "zero1") != "allreduce":
            # data parallel, SURVEY 8 f1: bucketed reduce-scatter over RCCL (transformer buckets under the backbone backward),
            # fused AdamW on the owned 1/world of the arena, all-gather of the updated parameters
            eng.backward_reduce_scatter(1.0 / world)
            eng.adamw_step_sharded(1e-5, 1e-5, 1e-4, step=i + 1)
        else:
            eng.backward_allreduce(1.0 / world)   # bucketed all-reduce, transformer range under the backbone backward; full AdamW
            eng.adamw_step(1e-5, 1e-5, 1e-4, step=i + 1)
        return out
    step.host_batch = lambda: tuple(torch.from_numpy(inp[k]).pin_memory() for k in ("image_u8", "qpos", "actions", "is_pad"))
    return eng, step


def train_record(cfg, B, dev, steps, warmup, train_prec=None):
    """extra.train_b64: the ACT training step (imitate_episodes.py:601-607) on one GPU, in the default fp32-grade arithmetic
    (fp32 storage, f16x3 products: bf16 products cannot meet the 1e-4 forward bar).  extra.train_b64_bf16 is BASELINE
    config 3 as written: the opt-in mode with ONE bf16 product per fp32 product in every GEMM of the step (fp32 accumulate,
    fp32 master weights and AdamW state; tests/test_gpu_training.py pins its distance to the reference gradients)."""
    import torch
    from actmi import lib as L
    eng, step = train_rig(cfg, B, dev, 777, 1, train_prec)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(warmup + i)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    # the kernel table comes from a pass of its own: the per-launch events of the in-library profiler cost ~4 % of the step
    L.profile_enable(True)
    for i in range(steps):
        step(warmup + steps + i)
    torch.cuda.synchronize(dev)
    L.profile_enable(False)
    prof, rows, gpu_ms = kernel_table(L.profile_report(), steps)
    dom, ach = dominant(prof)
    assert torch.isfinite(out["loss"]).all()
    # the same steps fed from the HOST the way train_bc feeds them: u8 NHWC batches in pinned memory, copied by
    # actmi.data.DevicePrefetcher on a side stream while the previous step computes (the reference moves f32 images, 4x the bytes,
    # synchronously in front of every step: imitate_episodes.py:529-532)
    from actmi.data import DevicePrefetcher
    host = step.host_batch()
    feed = DevicePrefetcher((host for _ in range(steps + 1)), device=dev)
    step(warmup + steps, next(feed))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(warmup + steps + 1 + i, next(feed))
    torch.cuda.synchronize(dev)
    dt_fed = time.perf_counter() - t0
    assert torch.isfinite(out["loss"]).all()
    del eng
    return {"per_gpu_batch": B, "steps": steps, "ms_per_step": dt / steps * 1e3, "samples_per_s": B * steps / dt,
            "host_fed": {"ms_per_step": dt_fed / steps * 1e3, "h2d_bytes_per_step": int(sum(t_.numel() * t_.element_size() for t_ in host)),
                         "note": "every step's batch (u8 NHWC frames, qpos, actions, is_pad) copied from pinned host memory by "
                                 "actmi.data.DevicePrefetcher on a side stream under the previous step"},
            "train_steps_per_s": steps / dt,
            "arithmetic": ("fp32 storage / accumulation / master weights / AdamW state, ONE bf16 MFMA product per fp32 product in every "
                           "GEMM and implicit-GEMM convolution of the step (opt-in train_prec=bf16: BASELINE config 3 as written; "
                           "the direct stem / layer1 kernels and the fused attention kernels, forward and backward, keep f16x3); dropout 0" if train_prec == "bf16" else
                           "fp32 storage and accumulation, f16x3 products forward and backward, fused AdamW; dropout 0 "
                           "(the fp32-grade default step; extra.train_b64_bf16 is the bf16 speed mode)"),
            "achieved_tflops_live": B * steps / dt * GFLOP_TRAIN_PER_SAMPLE_LIVE / 1e3,
            "dominant_kernel": dom["name"], "dominant_tflops": ach, "dominant_frac": ach / kernel_peak(dom["name"])[0],
            "dominant_share": dom["ms"] / gpu_ms,
            "top_kernels": [{"name": r["name"], "share": r["share"], "avg_us": r["avg_us"], "tflops": r["tflops"]} for r in rows[:6]]}


def diffusion_record(dev, B, iters, warmup):
    """extra.diffusion_b32 (BASELINE config 4): DiffusionPolicy batched inference, 3 cameras 480x640, prediction horizon 32,
    the fork's DDIM schedule (10 inference steps of a 50-step scheduler, imitate_episodes.py:104, policy.py:102-109 -- the
    config string's "100 denoise steps" is the training-step count of commands.txt:104, not what inference runs).  Restated
    robomimic / diffusers arithmetic: parity unpinned (oracle/diffusion_ref.py)."""
    import torch
    from actmi import weights as W
    from actmi.diffusion import DiffusionNet, generate_diffusion_state_dict
    cams = ["top", "left_wrist", "right_wrist"]
    net = DiffusionNet(cams, prediction_horizon=32, num_inference_timesteps=10, device=str(dev))
    net.load_state_dict(generate_diffusion_state_dict(net.spec, seed=0))
    img = torch.from_numpy(W.rand_u8(3, "dimg", (B, len(cams), 480, 640, 3))).to(dev)
    qpos = torch.zeros((B, 14), device=dev)
    noise = torch.randn((B, 32, 16), device=dev)
    for _ in range(warmup):
        out = net.forward_infer(qpos, img, noise=noise)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        out = net.forward_infer(qpos, img, noise=noise)
    torch.cuda.synchronize(dev)
    dt_eager = (time.perf_counter() - t0) / iters
    replay = net.capture_infer(B, img)                  # the whole query as one hipGraph (what DiffusionPolicy.__call__ runs)
    out_g = replay(qpos, img, noise=noise)
    torch.cuda.synchronize(dev)
    assert torch.equal(out_g, out), "graph replay differs from eager launches"
    t0 = time.perf_counter()
    for _ in range(iters * 3):
        out = replay(qpos, img, noise=noise)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / (iters * 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cond = net.obs_cond(qpos, img)
    e0.record()
    net.unet(noise, 45, cond)
    e1.record()
    torch.cuda.synchronize(dev)
    assert torch.isfinite(out).all()
    return {"per_gpu_batch": B, "cameras": len(cams), "prediction_horizon": 32, "ddim_steps": 10, "ms_per_query_batch": dt * 1e3,
            "policy_steps_per_s": B / dt, "unet_pass_ms_eager": e0.elapsed_time(e1), "ms_per_query_batch_eager": dt_eager * 1e3,
            "launch": "hipGraph replay of the whole query (eager: op-level C ABI calls from Python)",
            "parity": "unpinned: robomimic / diffusers restated from their published definitions (not importable offline)"}


def bench_train(args, cfg, B, ctx):
    """ACT training step (imitate_episodes.py:601-607): zero_grad, forward (CVAE + policy), L1+KL, backward, AdamW."""
    import torch
    from actmi import lib as L
    rank, world, dev, dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["dist"]
    log(f"rank {rank}/{world}: building training engine (batch {B})")
    eng, step = train_rig(cfg, B, dev, 1234 + rank, world, args.train_prec)
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    log("warm-up done")
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    # the kernel table comes from an untimed pass of its own: the in-library profiler brackets every launch with events (~4 % of the step)
    L.profile_enable(True)
    for i in range(args.steps):
        step(args.warmup + args.steps + i)
    torch.cuda.synchronize(dev)
    L.profile_enable(False)
    prof = L.profile_report()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out["loss"]).all()
    if rank != 0:
        return None
    prof, rows, gpu_ms = kernel_table(prof, args.steps)
    dom, ach = dominant(prof)
    value = world * B * args.steps / elapsed
    return {
        "metric": "ACT training samples/sec (fwd+bwd+AdamW, 4x480x640 cams, chunk=100)", "value": value,
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32 storage / accumulate / master weights, bf16 products (one bf16 MFMA per fp32 product)" if args.train_prec == "bf16"
                  else DTYPE["f32" if os.environ.get("ACTMI_GEMM_PREC", "f16x3").startswith("f3") else "f16x3"]), "data": "synthetic",
        "train_prec": args.train_prec or "f16x3",
        "config": {"workload": f"ACT training step, per-GPU batch {B}, 4 cams 480x640, hidden 512, ff 3200, dropout 0",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}" if world > 1 else "single"},
        "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": kernel_peak(dom["name"])[0],
                     "unit": "TFLOP/s", "frac": ach / kernel_peak(dom["name"])[0], "traffic": None,
                     "peak_is": kernel_peak(dom["name"])[1], "avg_launch_us": dom["ms"] * 1e3 / dom["count"]},
        "whole_step": {"gflop_per_sample_live": GFLOP_TRAIN_PER_SAMPLE_LIVE,
                       "achieved_tflops_live": value / world * GFLOP_TRAIN_PER_SAMPLE_LIVE / 1e3,
                       "gpu_kernel_ms_per_step": gpu_ms / args.steps},
        "kernels": [{k: r[k] for k in ("name", "launches_per_step", "avg_us", "share", "tflops")}
                    for r in (rows if os.environ.get("ACTMI_PROF_SHAPES") == "1" else rows[:16])],
    }


# --------------------------------------------------------------------------------------------------------------
# episode-sharded eval rollouts (BASELINE config 5)
# --------------------------------------------------------------------------------------------------------------
def bench_eval_shard(args, ctx):
    """sim_insertion_scripted: 50 episodes per GPU in lock-step, temporal ensembling, SyntheticEnv, one RCCL all-gather of
    (episode_return, highest_reward); value = policy steps/s over the whole job, rollout loop included (env stepping on
    host threads, u8 frames over PCIe every step, one D2H of the ensembled action per step)."""
    import tempfile
    import torch
    import imitate_episodes as IE
    from actmi.constants import SIM_TASK_CONFIGS
    from actmi import dist_utils
    rank, world, dev, dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["dist"]
    task = "sim_insertion_scripted"
    tc = SIM_TASK_CONFIGS[task]
    n_local = args.episodes_per_gpu
    n_total = n_local * world
    ep_len = args.episode_len or tc["episode_len"]
    iargs = {"task_name": task, "policy_class": "ACT", "ckpt_dir": tempfile.mkdtemp(prefix="actmi_eval_"), "batch_size": n_local,
             "max_batch": n_local, "seed": 0, "num_steps": 0, "lr": 1e-5, "kl_weight": 10, "chunk_size": 100, "hidden_dim": 512,
             "dim_feedforward": 3200, "temporal_agg": True, "eval_every": 0, "validate_every": 0, "save_every": 0,
             "synthetic_env": True}
    config = IE.build_config(iargs)
    config["episode_len"] = ep_len
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):             # the host side prints as the reference does: stdout carries the JSON line only
        policy = IE.make_policy("ACT", dict(config["policy_config"], training=False), device=str(dev))
    policy.eval()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sys.stderr):
        sr, avg_ret = IE.eval_bc(config, "policy_last.ckpt", save_episode=False, num_rollouts=n_total, policy=policy,
                                 max_parallel=n_local, verbose=(rank == 0))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    # which ranks contributed: gather each rank's id through the same collective path the metrics used
    ids = dist_utils.all_gather_rows(torch.full((1, 2), float(rank)), [1] * world)
    ranks_seen = sorted(int(x) for x in ids[:, 0].tolist())
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        return None
    steps_total = n_total * ep_len
    return {"metric": "policy steps/sec in episode-sharded eval rollouts (sim_insertion_scripted, 50 episodes/GPU)",
            "value": steps_total / elapsed, "unit": "policy steps/s", "n_gpus": world, "steps": ep_len, "warmup": 0,
            "ms_per_step": elapsed / ep_len * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE["f32" if os.environ.get("ACTMI_GEMM_PREC", "f16x3").startswith("f3") else "f16x3"], "data": "synthetic",
            "config": {"workload": f"{task}: {n_total} episodes sharded {n_local}/GPU, {ep_len} timesteps, 3 cams 480x640 u8, "
                                   "temporal_agg, SyntheticEnv stand-in (dm_control absent), one all-gather of "
                                   "(episode_return, highest_reward)",
                       "episodes": n_total, "per_gpu_batch": n_local, "parallelism": f"episode shards x{world}",
                       "collective": "all_gather over RCCL" if world > 1 else "none (1 rank)"},
            "ranks_seen": ranks_seen, "success_rate_synthetic_env": sr, "avg_return_synthetic_env": avg_ret,
            "seconds": elapsed}


def cpu_baseline(cfg, B, iters):
    """Time the oracle (CPU restatement of the reference forward, fp32, as written) on the host cores."""
    import torch
    from actmi import weights as W
    from oracle import act_ref as R
    # the GPU box gives a 1-GPU job a 16-core CPU share; more threads than that only thrash
    ncores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), int(os.environ.get("ACTMI_CPU_THREADS", "16")))
    torch.set_num_threads(ncores)
    log(f"cpu baseline: {ncores} threads")
    sd = {"model." + k: torch.from_numpy(v) for k, v in W.generate_state_dict(cfg, seed=0).items()}
    inp = W.generate_inputs(cfg, B, seed=1234)
    qpos = torch.from_numpy(inp["qpos"])
    image = torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"]))
    with torch.no_grad():
        R.policy_call(sd, cfg, qpos[:1], image[:1])               # warm-up
        log("cpu baseline: warm-up done")
        t0 = time.perf_counter()
        done = 0
        for _ in range(iters):
            R.policy_call(sd, cfg, qpos, image)
            done += 1
            log(f"cpu baseline: {done} forwards in {time.perf_counter() - t0:.1f} s")
            if time.perf_counter() - t0 > 30.0:
                break
        dt = time.perf_counter() - t0
        iters = done
    return {"value": B * iters / dt, "unit": "policy steps/s", "cores": ncores, "kind": "port",
            "sample": f"{iters} forwards of batch {B} (same workload, as-written 7 decoder layers) after 1 warm-up; "
                      f"torch {torch.__version__} CPU fp32, {ncores} threads"}


if __name__ == "__main__":
    main()
