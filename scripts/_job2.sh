cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $R/scripts/_lat.py > /tmp/kt.log 2>&1
python $R/scripts/timeline.py $(find /tmp/kt -name "*kernel_trace.csv") k_prepare 20 | cut -c1-110
