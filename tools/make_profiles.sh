#!/bin/bash
# All rocprofv3 summaries of one round, on the GPU box:  bash tools/make_profiles.sh r03
# Writes gpurun_out/profiles_<round>/<round>_*.{json,csv}; copy them into profiles/ afterwards (gpurun only brings
# gpurun_out/ back).  Every --pmc pass runs on its own, never together with a trace domain (tools/profile.sh).
set -u
RND=$1
R=$GRAFT_REPO_ROOT
W=gpurun_out/prof_$RND
O=$R/gpurun_out/profiles_$RND
mkdir -p $O
cd $R
# (a) the headline command without the extras: kernel trace + FETCH / WRITE / L2 passes of the cfg2 sweep
bash tools/profile.sh $W/cfg2 1 -- --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/log_cfg2.txt 2>&1
python3 tools/prof_summary.py $W/cfg2 $O/${RND}_cfg2_spmm "spmm_plan_kernel" cfg2-default > /dev/null
# (b) the default command (extras included): the kernels of SpMV / gather / merges / sortedness — WITHOUT the legs that run
# the same kernels on other (small) workloads, which are profiled on their own in (b2)
export MXGPU_BENCH_EXTRAS_SKIP=vignette_loop,vignette_dense_csc,spmm_short_rows,export_small_calls,spmm_zipf,skewed_rows_ops,cfg5_shard_skewed
bash tools/profile.sh $W/extras 1 -- --steps 10 --warmup 3 --no-cpu-baseline > $O/log_extras.txt 2>&1
unset MXGPU_BENCH_EXTRAS_SKIP
python3 tools/prof_summary_multi.py $W/extras $O/${RND}_extras extras-default spmv_flat_kernel slice_rows_kernel spmv_plan_kernel \
    gather_fused_kernel gather_count_kernel gather_copy_kernel "merge_count_kernel<64, false>" "merge_count_kernel<64, true>" \
    "merge_fill_kernel<64, 0" "merge_fill_kernel<64, 1" "merge_fill_kernel<64, 2" rows_sorted_tile_kernel stream_copy_kernel \
    csr_by_dvec_kernel "drop_count_kernel<32, 0, double>" "drop_fill_kernel<32, 0, double>" > /dev/null
cp "$(ls -t $W/extras/trace/*/*_kernel_stats.csv | head -1)" $O/${RND}_extras_kernel_stats.csv
# (b2) the reference's published workload alone (dense 100 x 1e4 %*% CSC 1e4 x 1e4): the LDS-tile kernel (AUTO since round 5), the
# row-split kernel and its cursors; SQ / LDS / TA counter passes as well
export MXGPU_BENCH_EXTRAS_ONLY=vignette_dense_csc
bash tools/profile.sh $W/vignette 2 -- --steps 5 --warmup 2 --no-cpu-baseline > $O/log_vignette.txt 2>&1
unset MXGPU_BENCH_EXTRAS_ONLY
python3 tools/prof_summary_multi.py $W/vignette $O/${RND}_vignette_extras vignette-dense-csc spmm_tile_kernel "spmm_rowsplit_kernel<double, 2, 64, false" \
    rowsplit_cursors_kernel "spmm_rowwave_kernel<double, 2, false" spmm_slab_kernel spmm_plan_kernel > /dev/null
cp "$(ls -t $W/vignette/trace/*/*_kernel_stats.csv | head -1)" $O/${RND}_vignette_kernel_stats.csv
# (c) configs[4]'s per-GPU shard on this GPU
bash tools/profile.sh $W/cfg5 1 -- --config cfg5 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/log_cfg5.txt 2>&1
python3 tools/prof_summary.py $W/cfg5 $O/${RND}_cfg5_shard "spmm_plan_kernel" cfg5-shard-default > /dev/null
# (d) the bench lines themselves (un-profiled runs: profiled passes clock lower).  The summaries just made go into this
# copy's profiles/ first, so that the lines quote THIS build's counters (bench.py refuses summaries whose kernel sources
# hash differently from the tree it runs in)
cp $O/${RND}_*.json $O/${RND}_*.csv $R/profiles/
python3 bench.py --steps 20 --warmup 5 > $O/${RND}_bench_n1.json 2> $O/log_bench.txt
cp gpurun_out/bench_extras.json $O/${RND}_bench_n1_full.json          # (the line is the <= 4 KB headline; every extra with its notes is here)
python3 bench.py --config cfg5 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/${RND}_cfg5_shard_bench_n1.json 2>> $O/log_bench.txt
python3 bench.py --config cfg5-full > $O/${RND}_cfg5_full_bench.json 2> $O/log_cfg5_full.txt
cp gpurun_out/bench_cfg5_full.json $O/${RND}_cfg5_full_bench_full.json
# (e) configs[4] whole: kernel trace of the same command
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$W/cfg5_full/trace -- python3 $R/bench.py --config cfg5-full > $O/log_cfg5_full_trace.txt 2>&1
cd $R
cp "$(ls -t $W/cfg5_full/trace/*/*_kernel_stats.csv | head -1)" $O/${RND}_cfg5_full_kernel_stats.csv
ls -la $O
