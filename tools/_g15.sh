O=gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/host_path_probe.py --gib 8
for mb in 64 512; do echo "chunk $mb MiB"; CW_HOST_CHUNK_MB=$mb python tools/host_path_probe.py --gib 8 --passes 2; done
rocprofv3 --kernel-trace --memory-copy-trace -d $O/hostpath_trace -o p -f csv -- python3 tools/host_path_probe.py --gib 4 --passes 2 > $O/hostpath_trace.log 2>&1
cat $O/hostpath_trace.log | tail -3; ls $O/hostpath_trace
