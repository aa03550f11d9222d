cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for shape in "fwd 32 5 18 512 512 5:2" "fwd 32 5 18 512 512 5:0" "fwd 32 5 18 256 256 5:2" "dgrad 32 5 18 512 512 5:0" "fwd 32 10 37 256 256 5:2"; do
  a=${shape%%:*}; pad=${shape##*:}
  for p in 0 8; do
    PAD=$pad REPS=20 VCA_PROBE=$p timeout -k 10 60 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hq -- python3 $R/tools/probes/conv_probe.py $a 1 mixed > /dev/null 2>&1 || exit 1
    echo "$a pad $pad probe $p: $(grep halo $R/gpurun_out/hq/*/*kernel_stats.csv | awk -F, '{print $(NF-5), $(NF-4)}')"
    rm -rf $R/gpurun_out/hq
  done
done
