// cgic_launch.hip -- host-side launcher for several captured hipGraphs at once (pipeline.LaneStream: one graph per lane).
//
// A lane's work is one hipGraphLaunch on its own stream; from Python the four launches of a submit are four interpreter round
// trips (stream context manager + binding + hipGraphLaunch, ~25-30 us each), so lane 3 starts ~100 us after lane 0 -- a
// visible share of a 20-step window of ~800 us.  Here the launches are one C call, back to back on the calling thread.
// (Round 3 also had one persistent worker thread per lane: the runtime serialises the launches, the window got no shorter,
// and the pool was not reentrant -- removed in round 4, NOTES.md.)
#include "cgic_common.h"

extern "C" int cgic_launch_graphs(void *const *graph_execs, void *const *streams, int n)
{
    CGIC_REQUIRE(n >= 0 && (n == 0 || (graph_execs && streams)), CGIC_ERR_INVALID, "launch_graphs: NULL argument");
    for (int i = 0; i < n; ++i) CGIC_HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_execs[i], (hipStream_t)streams[i]));
    return CGIC_OK;
}
