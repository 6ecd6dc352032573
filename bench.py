#!/usr/bin/env python3
"""Headline benchmark: ACT policy steps/sec on synthetic 4-camera 480x640 batches (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one batched policy query through the C-ABI HIP path: u8 NHWC images already resident in HBM ->
multi-camera ResNet18 -> DETR encoder/decoder -> a_hat [B,100,16] -> temporal-ensemble reduction.  Weights are
random-init of the reference architecture (no network for checkpoints), inputs synthetic; work per step is
independent of the values.  N > 1 runs N independent replicas on disjoint batches (episodes shard embarrassingly;
the path has no data-path collective) and reports the whole-job aggregate with max-over-ranks timing.

Rank 0 prints ONE JSON line with the driver's contract plus:
  roofline     dominant kernel (the fp32 MFMA GEMM/implicit-conv instantiation with the largest total time):
               achieved = algorithmic FLOPs of its launches / their summed duration, measured with HIP events
               on the launch stream inside the timed region; peak = 157.3 TFLOP/s (fp32 matrix, MI355X_MICROARCH.md)
  cpu_baseline the CPU oracle (torch fp32 restatement of the reference, as written: all 7 decoder layers) timed on
               this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "act-plus-plus_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_MATRIX_TFLOPS = 157.3          # MI355X_MICROARCH.md, chip-level parameters
PEAK_FP16_MATRIX_TFLOPS = 16 * 157.3     # dense F16/BF16 MFMA = 16x the f32 MFMA rate (same guide, matrix-core table)


ARITH = {
    "f16x3": "fp32 in / fp32 out / fp32 accumulate; GEMM and conv products on the fp16 matrix pipe with both operands "
             "split exactly into two fp16 pieces (3 MFMA products per fp32 product, fp32-grade parity: GEMMs, 3x3/1x1 "
             "convolutions, attention); conv1 stem on the native fp32 MFMA; LayerNorm / softmax / pooling fp32 VALU",
    "f32": "fp32 everywhere, native fp32 MFMA (v_mfma_f32_32x32x2_f32)",
}


def kernel_peak(name):
    """Dense MFMA peak, in ALGORITHMIC fp32 FLOP/s, of the arithmetic a kernel uses.  gemm_f16x3_* forms every fp32
    product from three fp16 MFMA products (exact two-piece fp16 split of both operands), so its ceiling is a third of
    the fp16 matrix peak; everything else runs the native fp32 MFMA."""
    if name.startswith("gemm_f16x3") or name.startswith("attn_f16x3"):
        return PEAK_FP16_MATRIX_TFLOPS / 3.0, "f16 MFMA dense peak / 3 products per fp32 product"
    return PEAK_FP32_MATRIX_TFLOPS, "f32 MFMA dense peak"
GFLOP_PER_SAMPLE_LIVE = 145.4            # work that influences the output (decoder layer 0 only)
GFLOP_PER_SAMPLE_AS_WRITTEN = 160.4


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (metric is quoted at 8)")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer = headline policy-query metric; train = ACT training step (forward+backward+AdamW), fp32")
    ap.add_argument("--graph", action="store_true", default=True,
                    help="(default) replay the step as ONE captured hipGraph; per-kernel events then come from extra eager "
                         "steps after the timed region, since events cannot bracket kernels inside a graph")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch the step's kernels eagerly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from actmi.config import ACTConfig
    from actmi import weights as W
    from actmi import lib as L
    from actmi import ops
    from actmi.engine import ACTEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})",
              file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)          # "nccl" is RCCL on ROCm

    cfg = ACTConfig()                                           # C=4, 480x640, Q=100, D=512, F=3200, 4 enc + 7 dec
    B = args.batch
    if args.mode == "train":
        return bench_train(args, cfg, B, dev, rank, world, dist)
    log(f"rank {rank}/{world}: generating weights")
    eng = ACTEngine(cfg, max_batch=B, device=str(dev))
    eng.load_state_dict(W.generate_state_dict(cfg, seed=0))
    eng.finalize()
    log("engine ready")
    inp = W.generate_inputs(cfg, B, seed=1234 + rank)
    qpos = torch.from_numpy(inp["qpos"]).to(dev)
    image = torch.from_numpy(inp["image_u8"]).to(dev)           # resident in HBM before the timed region
    a_hat = torch.empty((B, cfg.num_queries, cfg.action_dim), dtype=torch.float32, device=dev)
    ens = ops.TemporalEnsemble(B, cfg.num_queries, cfg.action_dim, 0.01, dev)

    use_graph = args.graph
    replay = None
    if use_graph:
        try:
            replay = eng.capture_infer(B, with_ensemble=ens)     # forward + ensemble as ONE hipGraph launch
        except Exception as e:                                   # capture unsupported -> eager launches
            log(f"hipGraph capture failed ({e}); falling back to eager launches")
            use_graph = False

    def step():
        if use_graph:
            out, raw = replay(qpos, image)
            return raw
        eng.forward_infer(qpos, image, out=a_hat)
        return ens.step(a_hat)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    log("warm-up done")
    L.profile_enable(not use_graph)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    if use_graph:                      # per-kernel timing leg: the same number of steps, eager, outside the timed region
        L.profile_enable(True)
        for _ in range(args.steps):
            eng.forward_infer(qpos, image, out=a_hat)
            ens.step(a_hat)
        torch.cuda.synchronize()
    L.profile_enable(False)
    prof = L.profile_report()
    log(f"timed region done: {(t1 - t0) / args.steps * 1e3:.2f} ms/step")
    # per-step latency distribution (median + p10/p90, event-timed), after the timed region: each step
    # bracketed by events on the stream it runs on
    lat = None
    if rank == 0:
        n_lat = max(10, min(200, args.steps))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_lat)]
        for e0, e1 in evs:
            e0.record()
            step()
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        lat = {"n": n_lat, "p10": ts[int(0.1 * (n_lat - 1))], "p50": ts[n_lat // 2], "p90": ts[int(0.9 * (n_lat - 1))]}
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if use_graph:
        a_hat = replay.static[2]
    assert torch.isfinite(a_hat).all()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        # dominant kernel
        prof = [p for p in prof if p["ms"] > 0]
        prof.sort(key=lambda p: -p["ms"])
        dom = prof[0]
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        gpu_ms = sum(p["ms"] for p in prof)
        kernels = [{"name": p["name"], "launches_per_step": p["count"] / args.steps,
                    "avg_us": p["ms"] * 1e3 / p["count"], "share": p["ms"] / gpu_ms,
                    "tflops": (p["flops"] / (p["ms"] * 1e-3) / 1e12) if p["flops"] else None,
                    "gbps": p["bytes"] / (p["ms"] * 1e-3) / 1e9} for p in prof]
        # HBM traffic per launch of the dominant kernel: measured in separate rocprofv3 --pmc passes (FETCH_SIZE,
        # WRITE_SIZE; gfx950 correction applied) of this same command and committed under profiles/; null if absent
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_j_traffic.json")))
            traffic = tj["kernels"][dom["name"]]["traffic_bytes_per_launch"] if B == 8 else None
        except Exception:
            traffic = None
        out = {
            "metric": "policy steps/sec (4x480x640 cams, chunk=100, bs=8)",
            "value": value, "unit": "policy steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": ARITH.get(os.environ.get("ACTMI_GEMM_PREC", "f16x3"), ARITH["f16x3"]),
            "config": {"workload": "ACT eval policy query: 4 cams 480x640 u8, chunk 100, hidden 512, ff 3200, "
                                   "4 enc + 7 dec layers (layer 0 live), per-GPU batch %d, + temporal ensemble" % B,
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"replicas x{world}",
                       "launch": "hipGraph replay" if use_graph else "eager"},
            "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": kernel_peak(dom["name"])[0],
                         "unit": "TFLOP/s", "frac": ach / kernel_peak(dom["name"])[0], "traffic": traffic,
                         "peak_is": kernel_peak(dom["name"])[1],
                         "frac_of_native_fp32_mfma_peak": ach / PEAK_FP32_MATRIX_TFLOPS,
                         "avg_launch_us": dom["ms"] * 1e3 / dom["count"], "launches_per_step": dom["count"] / args.steps,
                         "flop_per_launch": dom["flops"] / dom["count"],
                         "algorithmic_bytes_per_launch": dom["bytes"] / dom["count"]},
            "whole_step": {"gflop_per_sample_live": GFLOP_PER_SAMPLE_LIVE,
                           "achieved_tflops_live": value / world * GFLOP_PER_SAMPLE_LIVE / 1e3,
                           "frac_of_fp32_matrix_peak": value / world * GFLOP_PER_SAMPLE_LIVE / 1e3 / PEAK_FP32_MATRIX_TFLOPS,
                           "gpu_kernel_ms_per_step": gpu_ms / args.steps},
            "kernels": kernels,
            "step_latency_ms": lat,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, B, args.cpu_iters)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def bench_train(args, cfg, B, dev, rank, world, dist):
    """ACT training step (imitate_episodes.py:601-607): zero_grad, forward (CVAE + policy), L1+KL, backward, AdamW."""
    import torch
    from actmi import weights as W
    from actmi import lib as L
    from actmi.engine import ACTEngine
    log(f"rank {rank}/{world}: building training engine (batch {B})")
    eng = ACTEngine(cfg, max_batch=B, device=str(dev), training=True)
    eng.load_state_dict(W.generate_state_dict(cfg, seed=0))
    eng.finalize()
    inp = W.generate_inputs(cfg, B, seed=1234 + rank, with_actions=True)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}

    def step(i):
        eng.zero_grad()
        out = eng.forward_train(t["qpos"], t["image_u8"], t["actions"], t["is_pad"], eps=t["eps"])
        eng.backward(1.0 / world)
        if world > 1:
            eng.allreduce_grads()          # data-parallel training: bucketed gradient all-reduce over RCCL / xGMI
        eng.adamw_step(1e-5, 1e-5, 1e-4, step=i + 1)
        return out

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    log("warm-up done")
    L.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    L.profile_enable(False)
    prof = [p for p in L.profile_report() if p["ms"] > 0]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out["loss"]).all()
    if rank == 0:
        prof.sort(key=lambda p: -p["ms"])
        gpu_ms = sum(p["ms"] for p in prof)
        dom = prof[0]
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        gflop_live = 447.0                                      # fwd 149 incl. CVAE + bwd ~2x, per sample
        value = world * B * args.steps / elapsed
        print(json.dumps({
            "metric": "ACT training samples/sec (fwd+bwd+AdamW, 4x480x640 cams, chunk=100)", "value": value,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ACT training step, per-GPU batch {B}, 4 cams 480x640, hidden 512, ff 3200, dropout 0",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": kernel_peak(dom["name"])[0],
                         "unit": "TFLOP/s", "frac": ach / kernel_peak(dom["name"])[0], "traffic": None,
                         "peak_is": kernel_peak(dom["name"])[1],
                         "avg_launch_us": dom["ms"] * 1e3 / dom["count"]},
            "whole_step": {"gflop_per_sample_live": gflop_live, "achieved_tflops_live": value / world * gflop_live / 1e3,
                           "frac_of_fp32_matrix_peak": value / world * gflop_live / 1e3 / PEAK_FP32_MATRIX_TFLOPS,
                           "gpu_kernel_ms_per_step": gpu_ms / args.steps},
            "kernels": [{"name": p["name"], "launches_per_step": p["count"] / args.steps, "avg_us": p["ms"] * 1e3 / p["count"],
                         "share": p["ms"] / gpu_ms, "tflops": (p["flops"] / (p["ms"] * 1e-3) / 1e12) if p["flops"] else None}
                        for p in prof[:16]],
        }))
    if world > 1:
        dist.destroy_process_group()


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
