"""bench.py host logic without a GPU: the self-launcher refuses cleanly (non-zero exit, nothing run) when the box has
fewer GPUs than --gpus asks for, and the per-shape table parser reads the profiler's class names."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_exits_nonzero_without_enough_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU" in r.stderr and not r.stdout.strip().startswith("{")


def test_shape_table_and_peaks():
    import bench
    prof = [{"name": "gemm_f16x3_kernel<128,128,64,64,1,0>[M=2400,N=512,K=4608,g=4,sk=1,wgs=304]", "count": 60, "ms": 12.0,
             "flops": 60 * 2.0 * 2400 * 512 * 4608 * 4, "bytes": 1.0},
            {"name": "layernorm_kernel<2>", "count": 10, "ms": 0.1, "flops": 0.0, "bytes": 1.0}]
    rows = bench.shape_table(prof, 20)
    assert len(rows) == 1 and rows[0]["workgroups"] == 304 and rows[0]["K"] == 4608 and rows[0]["launches_per_step"] == 3
    assert abs(rows[0]["avg_us"] - 200.0) < 1e-9
    assert abs(bench.kernel_peak("gemm_f16x3_kernel<..>")[0] - 16 * 157.3 / 3) < 1e-9
    assert bench.kernel_peak("gemm_f32_kernel<..>")[0] == 157.3
