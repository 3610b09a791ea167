mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash tools/gpu_prof.sh r4_v1_bench python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -14
bash tools/gpu_prof.sh r4_v1_render python tools/bench_render_single.py 2>&1 | tail -8
G3_BENCH_BACKEND=gloo G3_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_bench8_share.log 2>&1; echo "bench8 share rc=$?"; grep '^{' gpurun_out/r4_bench8_share.log | cut -c1-1500
timeout 1500 python tools/psnr_vs_oracle.py --random_init --num_steps 1 --video_save_folder gpurun_out/psnr_out --json gpurun_out/r4_psnr_fullsize.json > gpurun_out/r4_psnr_fullsize.log 2>&1; echo "psnr rc=$?"; tail -8 gpurun_out/r4_psnr_fullsize.log
rm -rf gpurun_out/psnr_out gpurun_out/prof_r4_v1_bench gpurun_out/prof_r4_v1_render
