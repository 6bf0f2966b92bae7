/*
 * multiviewnative.h -- drop-in C ABI of the MI355X-native multi-view deconvolution library.
 *
 * Binary compatible with psteinb/libmultiviewnative's inc/multiviewnative.h for the GPU hot
 * path, so Fiji/SPIM_Registration (JNA) can load this library unchanged.  Each declaration
 * names the reference declaration it replaces.  Struct layout (x86-64 SysV == JNA default):
 * view_data = 8 pointers = 64 bytes; workspace = {view_data* @0, unsigned short @8,
 * double @16, float @24, int @28} = 32 bytes, passed BY VALUE.
 *
 * Differences in behaviour, all deliberate:
 *  - errors never terminate the host process (the reference calls exit(1) inside the JVM,
 *    inc/cuda_helpers.cuh:17-24): the boundary is noexcept, a diagnostic goes to stderr and
 *    the in/out buffer is left untouched;
 *  - inplace_gpu_deconvolve follows the reference GPU entry's zero_padd policy
 *    (src/multiviewnative.cu:26-27,128, inc/padd_utils.h:121-138, src/gpu_deconvolve_methods.cuh:
 *    366-449,537-549): every stack is embedded in a zero volume of extent >= image + kernel - 1
 *    at offset (kernel - 1)/2, the loop runs cyclically on that volume and psi is cropped on
 *    exit -- so a block does not wrap PSF energy across its borders.  Two additions: the padded
 *    extents grow to FFT-friendly 2^a 3^b 5^c 7^d sizes (542 = 2 * 271 -> 576), and the quotient is
 *    0 wherever a view voxel is exactly 0 (in the added zeros the reference's 0 * 1/0 would be
 *    NaN).  mvn_set_pad_mode("zero_exact") (mvn_engine_api.h) or MVN_PAD_MODE=zero_exact keeps
 *    exactly image + kernel - 1 without the guard; "none" selects the reference CPU path's
 *    no_padd instead (cyclic on exactly image_dims_, inc/cpu_convolve.h:22-26), which is the
 *    parity target of the oracle tests;
 *  - inplace_gpu_convolution and the legacy convolution entry points are cyclic on exactly the
 *    given dims with the kernel centre on the origin, as in the reference (as_is_padding,
 *    src/multiviewnative.cu:29-30,39-40,58-75);
 *  - the CPU entry points (inplace_cpu_deconvolve / inplace_cpu_convolution,
 *    inc/multiviewnative.h:43-51) are NOT exported by the product library: this library has
 *    no CPU fallback.  Their restatement lives in oracle/ as test infrastructure.
 */
#ifndef MVN_AMD_MULTIVIEWNATIVE_H
#define MVN_AMD_MULTIVIEWNATIVE_H

#include <stddef.h>

#ifdef __cplusplus
#define MVN_API extern "C" __attribute__((visibility("default")))
#else
#define MVN_API __attribute__((visibility("default")))
#endif

typedef float imageType; /* inc/multiviewnative.h:4 */

/* inc/multiviewnative.h:15-26 */
struct view_data {
  imageType* image_;
  imageType* kernel1_;
  imageType* kernel2_;
  imageType* weights_;
  int* image_dims_;
  int* kernel1_dims_;
  int* kernel2_dims_;
  int* weights_dims_;
};

/* inc/multiviewnative.h:28-35 */
struct workspace {
  struct view_data* data_;
  unsigned short num_views_;
  double lambda_;
  float minValue_;
  int num_iterations_;
};

#ifndef __cplusplus
typedef struct view_data view_data;
typedef struct workspace workspace;
#endif

/* inc/multiviewnative.h:66-67 (impl. src/multiviewnative.cu:89-142): multi-view
 * Richardson-Lucy on the GPU.  psi: in/out, prod(data_[0].image_dims_) floats, host memory.
 * device < 0 selects a device automatically.
 * Several GPUs (the reference drives one): with the environment variable MVN_DEVICES=0,1,..  (read per call) the
 * call cuts the padded volume into slabs of dim0 planes, one per listed device, and sweeps them in the same
 * view-after-view order with a halo exchange per convolution - same result as on one device, `device` is then
 * ignored.  A call that cannot be cut that way (a PSF deeper than 33 planes, fewer planes per slab than half the
 * deepest PSF, an odd last extent) runs on one device as usual.  See INTEGRATION.md section 4. */
MVN_API void inplace_gpu_deconvolve(imageType* psi, struct workspace input, int device);

/* inc/multiviewnative.h:59-61 (impl. src/multiviewnative.cu:58-75): in-place cyclic
 * convolution of im with kernel (centre -> origin), result normalised. */
MVN_API void inplace_gpu_convolution(imageType* im, int* imDim, imageType* kernel,
                                     int* kernelDim, int device);

/* inc/multiviewnative.h:77-79 (impl. src/multiviewnative.cu:199-241): legacy name of the
 * same host-pointer convolution. */
MVN_API void convolution3DfftCUDAInPlace(imageType* im, int* imDim, imageType* kernel,
                                         int* kernelDim, int devCUDA);

/* inc/multiviewnative.h:81-85 (impl. src/multiviewnative.cu:243-319): DEVICE-pointer
 * variant.  _d_imCUDA must hold imSize + 2*imDim[0]*imDim[1] floats (the reference's in-place
 * r2c allocation, src/multiviewnative.cu:213-216); only the first imSize floats, dense
 * [imDim0][imDim1][imDim2], are read and written. */
MVN_API void convolution3DfftCUDAInPlace_core(imageType* _d_imCUDA, int* imDim,
                                              imageType* _d_kernelCUDA, int* kernelDim,
                                              int devCUDA);

/* inc/multiviewnative.h:87-88 (impl. src/multiviewnative.cu:321-353):
 * _output[i] = _input[i] * (1 / _output[i]) on host arrays. */
MVN_API void compute_quotient(imageType* _input, imageType* _output, size_t _size, int _device);

/* inc/multiviewnative.h:89-93 (impl. src/multiviewnative.cu:355-393): RL update of _image
 * (psi) from _integral and _weight on host arrays; _lambda > 0 selects Tikhonov. */
MVN_API void compute_final_values(imageType* _image, imageType* _integral, imageType* _weight,
                                  size_t _size, float _minValue, double _lambda, int _device);

/* inc/multiviewnative.h:94-96 (impl. src/multiviewnative.cu:395-506): ONE Richardson-Lucy step
 * on a single host stack, the legacy demo path: psi_0 = view = _input, kernel1 = _kernel,
 * kernel2 = 0.1 in every tap (extents of _kernel), weights = 1, minValue = 1e-4, lambda = 0;
 * both convolutions cyclic on _input_dims.  _output receives the updated estimate. */
MVN_API void iterate_fft_plain(imageType* _input, imageType* _kernel, imageType* _output,
                               int* _input_dims, int* _kernel_dims, int _device);

/* inc/multiviewnative.h:98-102 (impl. src/multiviewnative.cu:508-600): as iterate_fft_plain, with
 * the legacy Tikhonov update of inc/cuda_kernels.cuh:162-193: t = image * integral;
 * t = t > 0 ? float((sqrt(1 + 2 * float(lambda) * t) - 1) / float(lambda)) : minValue;
 * out = w * (max(minValue, t) - t) + t (w = 1).  _size is unused (as in the reference). */
MVN_API void iterate_fft_tikhonov(imageType* _input, imageType* _kernel, imageType* _output,
                                  int* _input_dims, int* _kernel_dims, size_t _size,
                                  float _minValue, double _lambda, int _device);

/* inc/multiviewnative.h:104-109 (impl. inc/cuda_helpers.cuh:70-136).  The "CUDA" names are
 * kept for ABI compatibility; they report HIP devices.  "Compute capability" is the gfx
 * target split as major = gfx / 10 (e.g. 95), minor = gfx % 10 for gfx950. */
MVN_API int selectDeviceWithHighestComputeCapability(void);
MVN_API int getCUDAcomputeCapabilityMinorVersion(int devCUDA);
MVN_API int getCUDAcomputeCapabilityMajorVersion(int devCUDA);
MVN_API int getNumDevicesCUDA(void);
MVN_API void getNameDeviceCUDA(int devCUDA, char* name); /* copies 256 bytes */
MVN_API long long int getMemDeviceCUDA(int devCUDA);

#endif
