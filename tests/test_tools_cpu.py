"""The evidence tools that turn committed raw profiles into the numbers DESIGN.md quotes still run on those files
(tools/timeline_stats.py on the kernel timelines of profiles/: DESIGN.md section 6, last paragraph)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stats(*files):
    cp = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'timeline_stats.py')] +
                        [os.path.join(ROOT, 'profiles', f) for f in files], capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, cp.stderr[-2000:]
    return cp.stdout


def _in_flight(out):
    line = [l for l in out.splitlines() if 'kernels in flight' in l][0]
    import re
    return {int(k): float(v) for k, v in re.findall(r'(\d+)\)?: np\.float64\(([0-9.]+)\)', line)} or \
           {int(k): float(v) for k, v in re.findall(r'(\d+): ([0-9.]+)', line)}


def test_three_plain_streams_keep_the_gpu_full():
    out = _stats('r06_timeline_s3_b6.csv.gz', 'r06_timeline_s1_b6.csv.gz')
    share = _in_flight(out)
    assert share.get(0, 0.0) < 0.05 and share.get(3, 0.0) > 0.6, share      # never idle; three kernels in flight most of the time
    # small launches wait behind other streams' persistent kernels: grid_bbox_kernel stretches by more than 20x
    row = [l for l in out.splitlines() if 'grid_bbox_kernel' in l][0].split()
    assert float(row[-1]) > 20.0, row


def test_four_contexts_on_quarters_run_four_kernels_side_by_side():
    out = _stats('r06_timeline_s4_b6.csv.gz', 'r06_timeline_s1_b6.csv.gz')
    share = _in_flight(out)
    assert share.get(4, 0.0) > 0.6 and share.get(0, 0.0) < 0.1, share
    # on its own quarter no small launch is among the long ones any more
    assert 'grid_bbox_kernel' not in out
