# A/B of product-library variants on one box: bench.py's timed loop only, interleaved
#   bash tools/lib_ab.sh out.txt name1 name2 ...   ("default" = lib/libmultiviewnative.so, else lib/libmultiviewnative_<name>.so)
cd $GRAFT_REPO_ROOT
out=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    so=libmultiviewnative_amd/lib/libmultiviewnative_$v.so
    [ "$v" = default ] && so=libmultiviewnative_amd/lib/libmultiviewnative.so
    echo -n "$v: " >> $out
    MVN_PRODUCT_SO=$so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-abi 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['roofline']['per_kernel'].items()})" >> $out
  done
done
