import cProfile, pstats, sys, os, io
sys.argv = ["weg_time.py", "guided"]
sys.path.insert(0, "/root/repo")
pr = cProfile.Profile()
src = open("/root/repo/tools/weg_time.py").read()
pr.enable()
exec(compile(src, "/root/repo/tools/weg_time.py", "exec"), {"__name__": "__main__", "__file__": "/root/repo/tools/weg_time.py"})
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
