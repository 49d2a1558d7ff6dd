cd $GRAFT_REPO_ROOT
for cfg in "64 32" "128 64" "256 128" "512 512"; do set -- $cfg; echo "== K1=$1 K3=$2"; BT_TRAV_STATS=1 BT_V2_K1=$1 BT_V2_K3=$2 timeout 300 python bench.py --workload c4 --steps 3 --warmup 1 --cpu-sample 0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('[bt trav]'): last=l
    if l.startswith('{'):
        d=json.loads(l); print(last.strip()); print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stages_ms'].items() if k.startswith('trav')})
"; done
echo "== c3"; BT_TRAV_STATS=1 timeout 300 python bench.py --workload c3 --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep "bt trav" | tail -1
