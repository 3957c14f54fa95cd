set -e
O=gpurun_out/r2; mkdir -p $O
L=$O/hostpath_corpus2.log; rm -f $L
echo "== corpus sha256mb+lzf 4096, chunk 256 MiB (the configuration whose totals varied)" >> $L
CW_HOST_CHUNK_MB=256 timeout -k 10 300 python tools/host_path_probe.py --data corpus --hash sha256mb --comp lzf --bs 4096 --passes 4 >> $L 2>&1
for mb in 64 512 1024; do
echo "== random skein+lz4 4096, chunk $mb MiB" >> $L
CW_HOST_CHUNK_MB=$mb timeout -k 10 300 python tools/host_path_probe.py --data random --hash skein --comp lz4 --bs 4096 --passes 3 >> $L 2>&1
echo "== random sha256mb+lzf 4096, chunk $mb MiB" >> $L
CW_HOST_CHUNK_MB=$mb timeout -k 10 300 python tools/host_path_probe.py --data random --hash sha256mb --comp lzf --bs 4096 --passes 3 >> $L 2>&1
done
grep -v amdgpu.ids $L
