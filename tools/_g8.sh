mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python tools/psnr_vs_oracle.py --random_init --num_steps 3 --foreground_masking --video_save_folder gpurun_out/psnr_out --json gpurun_out/r4_psnr_fullsize_3steps.json > gpurun_out/r4_psnr_fullsize_3steps.log 2>&1; echo "psnr rc=$?"; grep "oracle chain\|worst frame" gpurun_out/r4_psnr_fullsize_3steps.log
rm -rf gpurun_out/psnr_out
timeout 1500 python tools/video_wallclock_cli.py 2>&1 | tail -4
ls gpurun_out/*wallclock*cli* 2>/dev/null
