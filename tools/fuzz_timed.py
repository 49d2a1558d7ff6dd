import sys, os, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools")); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import fuzz_parity
t0 = time.time(); tot = {}
seed = int(sys.argv[1]); budget = float(sys.argv[2])
while time.time() - t0 < budget:
    st = fuzz_parity.run(50, seed, verbose=False, aux=False)
    for k, v in st.items(): tot[k] = tot.get(k, 0) + v
    seed += 50
    print(seed, tot, round(time.time() - t0), flush=True)
