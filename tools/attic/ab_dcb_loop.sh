mkdir -p gpurun_out/r04/ab1
for rep in 1 2 3; do
for v in uniform head; do
  echo "== $v rep $rep" >> gpurun_out/r04/ab1/ab.txt
  D377_LIB=$PWD/build/variants/$v.so timeout -k 10 200 python3 tools/size_sweep.py --sizes 1048576,4194304 --ops sqrt_ratio_zeta,scalar_mul_base,scalar_mul_var,encode_to_curve 2>&1 | grep "n=" >> gpurun_out/r04/ab1/ab.txt
done
done
cat gpurun_out/r04/ab1/ab.txt
