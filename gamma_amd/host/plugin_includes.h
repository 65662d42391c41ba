// plugin_includes.h -- the only place that differs between the standalone build of this
// repository and a build inside the Gamma source tree: with -DGAMMA_HIP_IN_TREE
// the plugins compile against Gamma's real headers.
#pragma once
#ifdef GAMMA_HIP_IN_TREE
#include "common/gamma_common_data.h"
#include "index/retrieval_model.h"
#include "table/field_range_index.h"
#include "util/utils.h"
#else
#include "json_lite.h"
#include "plugin_api.h"
#endif
