# round 6, call 11: does keeping the matrix core's running sum small (separate accumulators per K-block group, added by the VALU)
# reduce the bias of its truncating alignment?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c11
(cd tools/microbench && timeout 300 ./bf16x3 2>&1 | sed -n '/^2b/,/^3\./p' > ../../gpurun_out/c11/numerics.txt); cat gpurun_out/c11/numerics.txt
