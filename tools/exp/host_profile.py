"""Where the host time of a training step goes: cProfile over 30 eager steps of bench.py's train loop (python tools/exp/host_profile.py)."""
import os, sys, cProfile, pstats, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ['bench.py', '--steps', '30', '--warmup', '5', '--no-cpu-baseline', '--train-only']
import bench
pr = cProfile.Profile()
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    pr.enable()
    bench.main()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats(45)
