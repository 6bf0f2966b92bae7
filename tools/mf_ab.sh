# A/B of the fused middle pass probe variants (build/probe/mfp_*), same box, interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
for rep in 1 2; do
for v in "$@"; do
  for k in 31 5; do
    echo -n "$v k=$k: "
    MF_NOCHECK=1 timeout -k 10 120 build/probe/mfp_$v 512 256 $k 0 30 1 | grep "best"
  done
done
done
