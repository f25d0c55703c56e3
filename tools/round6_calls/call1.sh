# round 6, call 1: the decision microbenchmarks for the general-state walk (verdict item 1a) + the GPU suite as it stands
mkdir -p gpurun_out/c1
cd tools/microbench
timeout 300 ./mfma_shadow > ../../gpurun_out/c1/mfma_shadow.txt 2>&1
timeout 300 ./bf16x3 > ../../gpurun_out/c1/bf16x3.txt 2>&1
cd ../..
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1/gputests.txt 2>&1
tail -3 gpurun_out/c1/gputests.txt
cat gpurun_out/c1/mfma_shadow.txt gpurun_out/c1/bf16x3.txt
