// TEMPORARY: engine + pipeline entry points land in the next milestone.
#include "common.h"
#define NI(name) adas::set_error(name ": not implemented yet"); return ADAS_ERR_INVALID
extern "C" {
int adas_engine_create(const char*, int, int, adas_engine**) { NI("adas_engine_create"); }
int adas_engine_destroy(adas_engine*) { return ADAS_OK; }
int adas_engine_input_shape(const adas_engine*, int64_t*) { NI("adas_engine_input_shape"); }
int adas_engine_num_outputs(const adas_engine*) { return 0; }
int adas_engine_output_shape(const adas_engine*, int, int64_t*, int*) { NI("adas_engine_output_shape"); }
const char* adas_engine_output_name(const adas_engine*, int) { return ""; }
int adas_engine_infer_host(adas_engine*, const float*, int, float* const*) { NI("adas_engine_infer_host"); }
int adas_engine_infer_device(adas_engine*, const float*, int, void*) { NI("adas_engine_infer_device"); }
const float* adas_engine_output_device(const adas_engine*, int) { return nullptr; }
int adas_engine_stats(const adas_engine*, double*, double*, int*) { NI("adas_engine_stats"); }
int adas_engine_profile(adas_engine*, const float*, int, int, float*, int, int*) { NI("adas_engine_profile"); }
int adas_engine_layer_info(const adas_engine*, int, char*, int, double*, int*) { NI("adas_engine_layer_info"); }
int adas_engine_fetch_activation(adas_engine*, int, int, float*, int64_t*) { NI("adas_engine_fetch_activation"); }
int adas_pipeline_create(const adas_pipeline_desc*, adas_pipeline**) { NI("adas_pipeline_create"); }
int adas_pipeline_destroy(adas_pipeline*) { return ADAS_OK; }
int adas_pipeline_step(adas_pipeline*, const float*, const float*) { NI("adas_pipeline_step"); }
int adas_pipeline_sync(adas_pipeline*) { NI("adas_pipeline_sync"); }
int adas_pipeline_timings(adas_pipeline*, float*) { NI("adas_pipeline_timings"); }
}
