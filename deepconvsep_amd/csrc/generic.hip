// Generic build_ca path -- placeholder until the strided-conv / pooling kernels land.
#include "generic.h"

struct DcsGenericNet {
    int unused;
};

int dcs_generic_create(dcs_ctx*, const DcsGenericDims&, int, int, int, const std::vector<std::vector<float>>&,
                       DcsGenericNet**) {
    DCS_FAIL(DCS_EUNSUPPORTED, "this build only carries the DSD100/hiphop graph on the GPU");
}
void dcs_generic_destroy(DcsGenericNet* g) { delete g; }
int dcs_generic_forward(DcsGenericNet*, const float*, int64_t, int, int, float*) {
    DCS_FAIL(DCS_EUNSUPPORTED, "generic network path not built");
}
int dcs_generic_separate(DcsGenericNet*, dcs_stft*, const float*, int64_t, int, int, float, int, int, float*, float*,
                         float*, float*, int64_t, DcsBuffer*) {
    DCS_FAIL(DCS_EUNSUPPORTED, "generic network path not built");
}
