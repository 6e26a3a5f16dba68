cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|name)[: ]+\S+|\b(TCP|TCC|TA|SQ|GRBM|TD)_[A-Z0-9_a-z]+" | sort -u | head -400 > gpurun_out/counters.txt
wc -l gpurun_out/counters.txt
