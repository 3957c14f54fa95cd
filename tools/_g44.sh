set -e
O=gpurun_out/r2; mkdir -p $O
L=$O/hostpath_chunk.log; rm -f $L
for cfg in "corpus skein lz4" "corpus sha256mb lzf" "random skein lz4" "random sha256mb lzf"; do set -- $cfg
for mb in 64 512 1024; do
echo "== host path, $1, $2+$3 4096 B blocks, 8 GiB, chunk $mb MiB" >> $L
CW_HOST_CHUNK_MB=$mb timeout -k 10 300 python tools/host_path_probe.py --data $1 --hash $2 --comp $3 --bs 4096 --passes 3 2>&1 | tail -1 >> $L
done; done
grep -v amdgpu.ids $L | paste - - | sed 's/pass 2: //'
