for v in "$@"; do
  LFA_LIB_PATH=image-analysis_amd/variants/liblfa_$v.so timeout -k 10 120 python tools/kernel_time.py 2>&1 | grep -v "amdgpu.ids" || exit 1
done
