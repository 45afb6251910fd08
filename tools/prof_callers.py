import os, sys, runpy, cProfile, pstats
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
os.makedirs("build", exist_ok=True)
sys.argv = ["train_battle.py", "--n_round", "1", "--map_size", "1000"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "oracle/_ref/callers/train_battle.py"), run_name="__main__")
finally:
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("cumtime").print_stats(45)
