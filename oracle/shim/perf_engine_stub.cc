// Stand-in for /root/reference/src/core/perf_engine.cc, the one reference TU that does not
// compile against nlohmann-json 3.1.1 (it needs json::get_to, added in 3.2). The in-memory
// PerfEngine (header-only part) is untouched; only JSON persistence is unavailable in the oracle.
#include "core/perf_engine.h"
namespace infini {
REGISTER_CONSTRUCTOR(0, PerfRecordObj::from_json);
void PerfEngine::savePerfEngineData(std::string) { IT_TODO_HALT_MSG("perf JSON needs nlohmann-json>=3.2"); }
void PerfEngine::loadPerfEngineData(std::string) { IT_TODO_HALT_MSG("perf JSON needs nlohmann-json>=3.2"); }
} // namespace infini
