import sys,re
def patch(path, pairs):
    s=open(path).read()
    for a,b in pairs:
        n=s.count(a)
        if n!=1:
            print("MISMATCH",n,path,a[:80]); sys.exit(1)
        s=s.replace(a,b)
    open(path,'w').write(s)

patch('/root/repo/x264_vs2008_amd/csrc/frame_core.hip', [('''extern "C" int x264hip_ssd_frame_async(''','''// x264_adaptive_quant_frame (R/encoder/ratecontrol.c:231-249): fenc->f_qp_offset of every macroblock from its AC energy.  The
// float arithmetic is the reference's, operation for operation (fp32 subtract / add / multiply are exactly rounded on both sides;
// the library is built with -ffp-contract=off).
static __device__ const float d_log2_lut[128] = {
    0.00000, 0.01123, 0.02237, 0.03342, 0.04439, 0.05528, 0.06609, 0.07682, 0.08746, 0.09803, 0.10852, 0.11894, 0.12928, 0.13955, 0.14975, 0.15987,
    0.16993, 0.17991, 0.18982, 0.19967, 0.20945, 0.21917, 0.22882, 0.23840, 0.24793, 0.25739, 0.26679, 0.27612, 0.28540, 0.29462, 0.30378, 0.31288,
    0.32193, 0.33092, 0.33985, 0.34873, 0.35755, 0.36632, 0.37504, 0.38370, 0.39232, 0.40088, 0.40939, 0.41785, 0.42626, 0.43463, 0.44294, 0.45121,
    0.45943, 0.46761, 0.47573, 0.48382, 0.49185, 0.49985, 0.50779, 0.51570, 0.52356, 0.53138, 0.53916, 0.54689, 0.55459, 0.56224, 0.56986, 0.57743,
    0.58496, 0.59246, 0.59991, 0.60733, 0.61471, 0.62205, 0.62936, 0.63662, 0.64386, 0.65105, 0.65821, 0.66534, 0.67243, 0.67948, 0.68650, 0.69349,
    0.70044, 0.70736, 0.71425, 0.72110, 0.72792, 0.73471, 0.74147, 0.74819, 0.75489, 0.76155, 0.76818, 0.77479, 0.78136, 0.78790, 0.79442, 0.80090,
    0.80735, 0.81378, 0.82018, 0.82655, 0.83289, 0.83920, 0.84549, 0.85175, 0.85798, 0.86419, 0.87036, 0.87652, 0.88264, 0.88874, 0.89482, 0.90087,
    0.90689, 0.91289, 0.91886, 0.92481, 0.93074, 0.93664, 0.94251, 0.94837, 0.95420, 0.96000, 0.96578, 0.97154, 0.97728, 0.98299, 0.98868, 0.99435};
__global__ __launch_bounds__(256) void k_aq_offset(const int *__restrict__ energy, float *__restrict__ out, int n, float strength)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 e = (u32)energy[i];
    const int lz = __clz((int)e);
    out[i] = strength * (d_log2_lut[(e << lz >> 24) & 0x7f] - lz + 16.573f);
}
extern "C" int x264hip_adaptive_quant_frame(x264hip_frame_ctx *c, const x264hip_picture *pic, float aq_strength, int32_t *energy_dev, float *offset_dev)
{
    if (x264hip_aq_var_frame(c, pic, energy_dev)) return -1;
    const int n = c->d.mb_w * c->d.mb_h * c->batch;
    const float strength = aq_strength * 1.0397;                  // ratecontrol.c:235 (float * double, rounded to float)
    hipLaunchKernelGGL(k_aq_offset, dim3((n + 255) / 256), dim3(256), 0, c->stream, energy_dev, offset_dev, n, strength);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int x264hip_ssd_frame_async(''')])
patch('/root/repo/include/x264hip.h', [('''int x264hip_aq_var_frame(x264hip_frame_ctx *c, const x264hip_picture *pic, int32_t *out_dev);''','''int x264hip_aq_var_frame(x264hip_frame_ctx *c, const x264hip_picture *pic, int32_t *out_dev);
/* x264_adaptive_quant_frame (R/encoder/ratecontrol.c:231-249): fenc->f_qp_offset[batch][n_mb] (float, device) from the energies
 * above; aq_strength = param.rc.f_aq_strength.  energy_dev: scratch [batch][n_mb] int32. */
int x264hip_adaptive_quant_frame(x264hip_frame_ctx *c, const x264hip_picture *pic, float aq_strength, int32_t *energy_dev, float *offset_dev);''')])
print('ok')
