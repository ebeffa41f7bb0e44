// conv_ml.h -- the multi-layer persistent convolution launch (conv_ml.hip, round 5): host-side interface.
//
// One launch runs a SET of convolution layers of one network (3x3 stride-1 / stride-2 halo tiles and 1x1 pointwise tiles, the shapes the
// per-layer kernels conv_halo / conv_pw take) from a table of work items (layer, tile, channel block).  Persistent workgroups pull items
// from a ticket counter; an item first waits until the frames it reads are complete in every layer that produces them (per-layer,
// per-frame arrival counters in HBM), and announces itself on its own layer's counter when its stores have left the CU.  Dependent
// layers therefore overlap wherever different frames are at different depths, independent branches (the two Detect branches of a pyramid
// level, the Detect convs of one level against the neck layers of the next) run side by side, and the launch boundary between the small
// 40x40 / 20x20 layers of the YOLO graphs (34 of YOLOv8n's 52 launches, 15-25 us each, one wave of workgroups) disappears.
#pragma once
#include "kernels.h"
#include <string>
#include <vector>

namespace adas {

constexpr int ML_MAX_DEPS = 6;      // producer layers an item may wait for (after transitive reduction)
constexpr int ML_MAX_LAYERS = 64;   // layers per launch (dependency closures are 64-bit masks)
constexpr int ML_GROUP_MAX = 8;     // layers of one grouped launch (conv_halo_group_kernel's block table)

// Is this conv (as engine_run_op would launch it at batch a.n, planned onto `kernel` = CONV_HALO | CONV_PW) a shape the multi-layer
// kernel has a tile body for AND one launch_conv would send to conv_halo / conv_pw (not to conv_halo_rw / conv_s2p / conv_h8)?
bool ml_layer_supported(const ConvArgs& a, int kernel);

struct MlPlan;   // device tables of one launch (owned by the engine, valid for one batch size)

struct MlPlanInfo {
    int n_layers = 0, n_items = 0, frames = 0, grid = 0, order = 0;
    size_t lds = 0;
    std::vector<int> items_per_layer;              // by layer (launch order)
    std::vector<std::vector<int>> deps;            // by layer: producer layers waited for (indices into the launch's layer list)
    std::vector<std::vector<int>> targets;         // arrivals per frame that complete each of those producers
    std::vector<uint64_t> item_words;              // the item table as uploaded (low word: tile / chunk, high word: layer | cb << 8 | frame << 16)
};

// Builds the tables for `layers` (given in a valid execution order: every layer after the layers whose outputs it reads).
// Returns nullptr and a reason when the set cannot run as one launch (too many layers / dependencies, a 32-bit offset overflow ...).
MlPlan* ml_plan_create(const std::vector<ConvArgs>& layers, const std::vector<int>& kernels, int prec, std::string* why, MlPlanInfo* info = nullptr,
                       bool host_only = false);
void ml_plan_destroy(MlPlan* p);
// memset of the control block (ticket, error word, arrival counters) + the kernel, on `st` (both are captured by a stream capture)
hipError_t ml_launch(const MlPlan* p, hipStream_t st);
// the control block's error word of the LAST launch (synchronises the device): 0, or 1 + the index of the item whose wait timed out
int ml_plan_status(const MlPlan* p, unsigned* error_word, unsigned* head16 = nullptr);   // head16: the control block's first 16 words (phase counters of -DADAS_ML_PROF builds)

// ---- grouped launch of INDEPENDENT layers (the default path, ADAS_NO_GROUP=1 disables): 3x3 halo convs of one dependency level of a run of
// consecutive layers -- the two Detect branches of a pyramid level, the Detect convs of different levels -- as ONE plain launch: block ->
// (layer, tile, channel block) through a table, each block runs conv_halo's tile body once.  No workgroup waits for another, nothing is
// persistent: the small layers ride in the tail of the large ones and their launch boundaries disappear.  Bit-identical to conv_halo.
bool group_layer_supported(const ConvArgs& a, int kernel);   // a conv launch_conv sends to conv_halo with a body in the group kernel
// dependency level of every layer of a run (0: depends on nothing inside the run), from the views: RAW, WAR and WAW hazards are edges
std::vector<int> ml_levels(const std::vector<ConvArgs>& layers);
struct MlGroup;
MlGroup* ml_group_create(const std::vector<ConvArgs>& layers, int prec, std::string* why);   // the layers must be mutually independent
void ml_group_destroy(MlGroup* g);
hipError_t ml_group_launch(const MlGroup* g, hipStream_t st);

}  // namespace adas
