for c in 0 64 80 96 112 128 160 192 256; do
  echo "== chunk override $c"
  SLAB_WORLDS=8,4 SODA_HIP_CHUNK_ROWS=$c python tools/slab_cost.py jacobi2d 16384 16384 144,16 2>&1 | grep -v amdgpu
done
