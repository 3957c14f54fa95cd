set -e
O=gpurun_out/r2; mkdir -p $O
L=$O/hostpath_corpus.log; rm -f $L
for cfg in "skein512 lz4 65536" "skein lz4 4096" "sha256mb lzf 4096"; do set -- $cfg
for mb in default 256 1024 2048; do
echo "== host path, corpus, $1+$2 $3 B blocks, 8 GiB, chunk $mb MiB" >> $L
if [ $mb = default ]; then timeout -k 10 300 python tools/host_path_probe.py --data corpus --hash $1 --comp $2 --bs $3 --passes 3 >> $L 2>&1
else CW_HOST_CHUNK_MB=$mb timeout -k 10 300 python tools/host_path_probe.py --data corpus --hash $1 --comp $2 --bs $3 --passes 3 >> $L 2>&1; fi
done; done
grep -v amdgpu.ids $L
