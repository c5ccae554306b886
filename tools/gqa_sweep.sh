# development: A/B of two builds of the library on the grouped-query shapes (base = build_probe/libhydragen_base.so)
for rep in 1 2; do
 for lib in build_probe/libhydragen_base.so hydragen_amd/csrc/libhydragen_hip.so; do
   echo "rep=$rep $(basename $lib)"
   HYDRAGEN_HIP_LIB=$lib timeout 600 python tools/phase_bench.py --only "C" 2>&1 | grep "^| C[35]"
   HYDRAGEN_HIP_LIB=$lib timeout 600 python tools/phase_bench.py --only "paper" 2>&1 | grep "^| paper"
 done
done
