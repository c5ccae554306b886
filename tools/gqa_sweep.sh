# development: the grouped-query suffix kernel's rate against where K and V sit (whole-job C5 shape, S = 128), several process starts
export HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so
for rep in 1 2 3; do
 for gap in -1 0 256 4096 69632 2099200 16777216 34359296; do
   for h in 1 8; do
     r=$(HYD_GQA_HPW=$h timeout 300 python tools/kbench.py fused --B 2048 --Hq 64 --Hkv 8 --S 128 --smax 256 --kv-gap $gap --iters 30 2>&1 | grep fused | sed -E 's/.*S= *[0-9]+ +([0-9.]+) us.*/\1/')
     echo "rep=$rep gap=$gap hpw=$h us=$r"
   done
 done
done
