# round 6, call 2: numerics variants of the bf16 x 3 split (rounded pieces, order of the products) + the round's new GPU tests
mkdir -p gpurun_out/c2
(cd tools/microbench && timeout 600 ./bf16x3 > ../../gpurun_out/c2/bf16x3.txt 2>&1)
timeout 900 python -m pytest tests -m gpu -x -q -k "path_kernels_at_baseline or beta_category" > gpurun_out/c2/newtests.txt 2>&1
tail -15 gpurun_out/c2/newtests.txt
sed -n '/^2b/,/^3\./p' gpurun_out/c2/bf16x3.txt
