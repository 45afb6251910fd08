R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O3 -o /tmp/scatter_chunks tools/probe/scatter_chunks.hip && /tmp/scatter_chunks > $O/scatter_chunks.txt 2>&1
grep -v "grid  2048\|grid  8192" $O/scatter_chunks.txt | head -60
python tools/render_locality.py 2>&1 | grep -v amdgpu.ids | tee $O/render_locality.txt
