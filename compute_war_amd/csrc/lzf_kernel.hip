// lzf_kernel.hip -- placeholder translation unit: the LZF kernel lands in the next milestone.
#include "cw_device.h"
namespace cw {
hipError_t lzf_launch(const uint8_t *, size_t, size_t, size_t, uint8_t *, size_t, uint32_t *, hipStream_t) { return hipErrorNotSupported; }
} // namespace cw
