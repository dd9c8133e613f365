# where a tick of the piles scene goes (every tick exact): kernel trace at 147456 bodies (and 2304 with an argument)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_piles; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for side in ${1:-128}; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$side -- python3 $R/scripts/time_piles_small.py $side > $O/log$side.txt 2>&1; grep bodies $O/log$side.txt
python3 $R/scripts/trace_busy.py $O/k$side 220 | tee $O/busy$side.txt
done
