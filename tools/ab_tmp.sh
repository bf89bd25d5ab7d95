for v in "SHF_LIB=variants/halo9.so" "SHF_LIB=variants/halo11.so" "SHF_X=0" "SHF_X=0" "SHF_LIB=variants/halo11.so" "SHF_LIB=variants/halo9.so"; do
  echo "$v"; env $v python bench.py --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; k=r['kernel_ms_per_image']; print(j['value'], k['conv_mfma_f16x3_w4_kernel<true, 4, 3>'], k['conv_mfma_f16x3_w4_kernel<false, 4, 3>'], k['conv_mfma_f16x3_w4_kernel<true, 2, 3>'], k['conv_mfma_f16x3_kernel<128, false, 1, 3>'])"
done
