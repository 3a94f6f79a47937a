"""Developer aid: phase times of the bench workload with and without the fused quadratise+backward kernel."""
import sys, os, json, io, contextlib
sys.path.insert(0, os.getcwd())
import bench
for nf in (0, 1):
    sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "10"] + (["--no-fusion"] if nf else []) + sys.argv[1:]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    d = json.loads(buf.getvalue().strip().split("\n")[-1])
    print("no_fusion=%d ms_per_step %.4f" % (nf, d["ms_per_step"]), {k: round(v, 4) for k, v in d["phase_ms"].items()}, flush=True)
    sys.argv = sys.argv[:1] + sys.argv[4 + nf:]
