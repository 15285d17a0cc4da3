import cProfile, pstats, sys, io
sys.argv = ["soak.py"]
pr = cProfile.Profile()
pr.enable()
exec(open("scratch/soak.py").read())
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
