# development: head dim 256 on the two suffix kernels (ablation library)
export HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so
for rep in 1 2; do
 for impl in valu gqa; do
   for shape in "8 1 2048" "32 8 64" "16 2 1024"; do
   set -- $shape
   r=$(HYD_SUFFIX_IMPL=$impl timeout 300 python tools/kbench.py fused --D 256 --B $3 --Hq $1 --Hkv $2 --S 16,64,128,256 --iters 20 2>&1 | grep fused | sed -E 's/.*S= *([0-9]+) +([0-9.]+) us.*/\1:\2/' | tr '\n' ' ')
   echo "rep=$rep $impl D=256 B=$3 heads=$1/$2  $r"
   done
 done
done
