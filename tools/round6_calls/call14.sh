# round 6, call 14: the fixed-topology chain (DNA 500 x 20 000) with the path and its log-likelihood in one launch: the device
# timeline of a stretch of generations and the host's per-call times (MBAMD_STATS=1), with and without (MBAMD_NO_FUSE_PATH=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c14; export TMPDIR=/tmp
{
for v in "" 1; do
  echo "== MBAMD_NO_FUSE_PATH=$v: host per-call times over 20 000 generations"
  env ${v:+MBAMD_NO_FUSE_PATH=1} timeout 600 python tools/mcmc_stats.py 500 20000 20000 dynamic fixed 2>&1 | grep -v "^\[mbamd\]     (pars"
done
echo "== timeline (fused)"
bash tools/prof_mcmc.sh 2>&1 | tail -60
} > gpurun_out/c14/mcmc_fixed.txt 2>&1
tail -90 gpurun_out/c14/mcmc_fixed.txt
