# development: A/B of two builds of the library on the grouped-query shapes (run through gpurun from the repo root).
#   cp hydragen_amd/csrc/libhydragen_hip.so build_probe/libhydragen_base.so   # before a change
#   ... change, rebuild ...
#   bash tools/gqa_sweep.sh
for rep in 1 2; do
 for lib in build_probe/libhydragen_base.so hydragen_amd/csrc/libhydragen_hip.so; do
   for shape in "64 8 2048" "8 1 2048" "32 8 64" "8 1 32" "16 2 2048" "32 4 2048"; do
   set -- $shape
   r=$(HYDRAGEN_HIP_LIB=$lib timeout 300 python tools/kbench.py fused --B $3 --Hq $1 --Hkv $2 --S 16,64,128,256 --iters 30 2>&1 | grep fused | sed -E 's/.*S= *([0-9]+) +([0-9.]+) us.*/\1:\2/' | tr '\n' ' ')
   echo "rep=$rep $(basename $lib) B=$3 heads=$1/$2  $r"
   done
 done
done
