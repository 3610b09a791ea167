"""gen3c_amd - MI355X-native GEN3C-Cosmos-7B denoising path (HIP kernels behind a C ABI + Python host mirror).

Importing the package is cheap and GPU-free; the HIP library is loaded on first use (gen3c_amd._lib.load) and there is
no CPU fallback."""
__version__ = "0.1.0"
