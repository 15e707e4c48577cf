// Temporary stand-ins until inter_mfma.hip / intra_mfma.hip land: report "not covered" so the
// dispatcher in c_api.hip always takes the generic path.
#include "conv_internal.h"
namespace epn {
bool intra_uses_mfma(int, int, int, int) { return false; }
int launch_inter_tables_mfma(const epn_inter_desc *, const float *, float *, float *, hipStream_t) { return EPN_EINVAL; }
int launch_inter_fwd_mfma(const epn_inter_desc *, const float *, const float *, const float *, const float *, float *, hipStream_t) { return EPN_EINVAL; }
int launch_inter_bwd_data_mfma(const epn_inter_desc *, const float *, const float *, const float *, const float *, float *, hipStream_t) { return EPN_EINVAL; }
int launch_inter_bwd_weight_mfma(const epn_inter_desc *, const float *, const float *, const float *, const float *, float *, hipStream_t) { return EPN_EINVAL; }
int launch_intra_fwd_mfma(const float *, const int32_t *, const float *, int, int, int, int, int, int, float *, hipStream_t) { return EPN_EINVAL; }
int launch_intra_bwd_data_mfma(const float *, const int32_t *, const float *, int, int, int, int, int, int, float *, hipStream_t) { return EPN_EINVAL; }
int launch_intra_bwd_weight_mfma(const float *, const float *, const int32_t *, int, int, int, int, int, int, float *, hipStream_t) { return EPN_EINVAL; }
}
namespace epn { bool inter_mfma_available() { return false; } }
