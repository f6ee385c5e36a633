// extern "C" entry into the REFERENCE's own ROIAlign arithmetic (test infrastructure only, built by oracle/build_ref.sh).
// This file is appended, at build time, behind lines 1-219 of /root/reference/pysgg/csrc/cpu/ROIAlign_cpu.cpp (the pure C++
// kernel templates PreCalc / pre_calc_for_bilinear_interpolate / ROIAlignForward_cpu_kernel; the ATen wrapper behind them
// targets torch 1.4's dispatch macros and is left out), which are compiled unmodified against the installed torch headers.
extern "C" void veto_ref_roi_align_forward(const float* feat, int channels, int height, int width, const float* rois, int n_rois,
                                           float spatial_scale, int pooled, int sampling_ratio, float* out) {
  ROIAlignForward_cpu_kernel<float>(n_rois * channels * pooled * pooled, feat, spatial_scale, channels, height, width, pooled,
                                    pooled, sampling_ratio, rois, out);
}
