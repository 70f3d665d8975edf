// cgic_launch.hip -- host-side launcher for several captured hipGraphs at once (pipeline.LaneStream: one graph per lane).
//
// A lane's work is one hipGraphLaunch on its own stream; from Python the four launches of a submit are four interpreter round
// trips (stream context manager + binding + hipGraphLaunch, ~25-30 us each), so lane 3 starts ~100 us after lane 0 -- a
// visible share of a 20-step window of ~800 us.  Here the launches are one C call: back to back on the calling thread, or
// (threads > 1) each on its own persistent worker thread so that the runtime's per-launch work overlaps.
#include "cgic_common.h"

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace cgic {

struct LaunchJob {
    hipGraphExec_t exec;
    hipStream_t stream;
};

class LaunchPool {
  public:
    ~LaunchPool()
    {
        {
            std::lock_guard<std::mutex> l(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    // lane k's jobs (in order) go to worker k; returns the first error
    hipError_t run(const std::vector<std::vector<LaunchJob>> &lanes, int device)
    {
        std::unique_lock<std::mutex> l(mu_);
        while ((int)th_.size() < (int)lanes.size()) {
            const int id = (int)th_.size();
            work_.emplace_back();
            th_.emplace_back([this, id] { loop(id); });
        }
        device_ = device;
        err_ = hipSuccess;
        pending_ = 0;
        for (size_t k = 0; k < lanes.size(); ++k) {
            if (lanes[k].empty()) continue;
            work_[k] = lanes[k];
            ++pending_;
        }
        ++gen_;
        cv_.notify_all();
        done_.wait(l, [this] { return pending_ == 0; });
        return err_;
    }

  private:
    void loop(int id)
    {
        int seen = 0;
        for (;;) {
            std::vector<LaunchJob> mine;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [&] { return stop_ || (gen_ != seen && !work_[id].empty()); });
                if (stop_) return;
                seen = gen_;
                mine.swap(work_[id]);
            }
            hipError_t e = hipSetDevice(device_);
            for (const LaunchJob &j : mine)
                if (e == hipSuccess) e = hipGraphLaunch(j.exec, j.stream);
            {
                std::lock_guard<std::mutex> l(mu_);
                if (e != hipSuccess && err_ == hipSuccess) err_ = e;
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    std::vector<std::vector<LaunchJob>> work_;
    int gen_ = 0, pending_ = 0, device_ = 0;
    bool stop_ = false;
    hipError_t err_ = hipSuccess;
};

}  // namespace cgic

using namespace cgic;

extern "C" int cgic_launch_graphs(void *const *graph_execs, void *const *streams, const int *lane_of, int n, int threads)
{
    CGIC_REQUIRE(n >= 0 && (n == 0 || (graph_execs && streams)), CGIC_ERR_INVALID, "launch_graphs: NULL argument");
    if (n == 0) return CGIC_OK;
    if (threads <= 1 || !lane_of) {
        for (int i = 0; i < n; ++i) CGIC_HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_execs[i], (hipStream_t)streams[i]));
        return CGIC_OK;
    }
    int lanes = 0;
    for (int i = 0; i < n; ++i) {
        CGIC_REQUIRE(lane_of[i] >= 0 && lane_of[i] < 64, CGIC_ERR_INVALID, "launch_graphs: lane %d outside [0, 64)", lane_of[i]);
        lanes = lane_of[i] + 1 > lanes ? lane_of[i] + 1 : lanes;
    }
    std::vector<std::vector<LaunchJob>> jobs((size_t)lanes);
    for (int i = 0; i < n; ++i) jobs[(size_t)lane_of[i]].push_back(LaunchJob{(hipGraphExec_t)graph_execs[i], (hipStream_t)streams[i]});
    int dev = 0;
    CGIC_HIP_TRY(hipGetDevice(&dev));
    static LaunchPool pool;
    const hipError_t e = pool.run(jobs, dev);
    if (e != hipSuccess) return hip_fail(e, "hipGraphLaunch (worker thread)", __FILE__, __LINE__);
    return CGIC_OK;
}
