// Test helper (GPU): the reference-shaped classes from several host threads at once, each thread with its own DeviceSlot (round-5
// review, Weak 8: device, kept context and the live registry were unguarded process-wide statics).
//   slots_check <l.ppm> <r.ppm> <max_dis> <threads> <pairs per thread> <out prefix> [device list: e.g. 0,0,0]
// Thread t works on GPU devices[t % n] through `DeviceSlot slot(device, keep_context = true)`: odd threads make the slot current
// (DeviceSlot::Use) and construct `PreCSPC` with the reference's seven-argument signature; even threads pass the slot to the extended
// constructor of DevicePlaneCost.  Every pair: PreCSPC -> CSPatchMatch::PatchMatch(2, ..., use_pp) with seed 100 + t, maps written to
// <prefix>_<t>_{l,r}.pgm after the LAST pair; planes() / disparity() are read while the cost object lives.  The main thread meanwhile
// runs the same class through the process-wide default slot (seed 99 -> <prefix>_main_{l,r}.pgm).
#include <atomic>
#include <sstream>
#include <thread>

#include "cc/grd_cc.h"
#include "cs_patchmatch.h"
#include "plane_cost/pre_cs_pc.h"

static std::atomic<int> g_failed(0);

static void one_thread(const Mat &l, const Mat &r, int max_dis, int t, int pairs, int device, const std::string &prefix) {
  try {
    DeviceSlot slot(device, /*keep_context=*/true, /*shared_gpu=*/t >= 2);  // threads 2.. run the folded sweep (CSPM_OPT_SWEEP_FOLD): same maps
    GrdCC cc(device);
    for (int k = 0; k < pairs; ++k) {
      std::unique_ptr<IPlaneCost> pc;
      if (t % 2) {
        DeviceSlot::Use use(slot);
        pc.reset(new PreCSPC(l, r, max_dis, 35, 5, &cc, 0.3));
      } else {
        pc.reset(new DevicePlaneCost(l, r, max_dis, 35, 5, &cc, 0.3, &slot));
      }
      CSPatchMatch pm(l, r, max_dis, 4);
      pm.set_seed(100 + t);
      pm.PatchMatch(2, pc.get(), t % 3 == 0);
      std::vector<double> d;
      pm.disparity(kLeft, &d);  // borrows the cost object's context: the registry lookup races with the other threads' adopt / disown
      if (d.size() != (size_t)l.rows * l.cols) ++g_failed;
      if (k == pairs - 1) {
        std::ostringstream a, b;
        a << prefix << "_" << t << "_l.pgm";
        b << prefix << "_" << t << "_r.pgm";
        if (!imwrite(a.str(), pm.dis(kLeft)) || !imwrite(b.str(), pm.dis(kRight))) ++g_failed;
      }
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "thread %d: %s\n", t, e.what());
    ++g_failed;
  }
}

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  Mat l = imread(argv[1]), r = imread(argv[2]);
  if (!l.data || !r.data) return 3;
  const int max_dis = std::atoi(argv[3]), nthreads = std::atoi(argv[4]), pairs = std::atoi(argv[5]);
  const std::string prefix = argv[6];
  std::vector<int> devices;
  if (argc > 7) {
    std::istringstream is(argv[7]);
    std::string tok;
    while (std::getline(is, tok, ',')) devices.push_back(std::atoi(tok.c_str()));
  }
  if (devices.empty()) devices.push_back(0);
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t) th.emplace_back(one_thread, std::cref(l), std::cref(r), max_dis, t, pairs, devices[t % devices.size()], prefix);
  try {  // the default slot, from the main thread, while the workers run
    GrdCC cc;
    for (int k = 0; k < pairs; ++k) {
      PreCSPC pc(l, r, max_dis, 35, 5, &cc, 0.3);
      CSPatchMatch pm(l, r, max_dis, 4);
      pm.set_seed(99);
      pm.PatchMatch(2, &pc, false);
      if (k == pairs - 1 && !(imwrite(prefix + "_main_l.pgm", pm.dis(kLeft)) && imwrite(prefix + "_main_r.pgm", pm.dis(kRight)))) ++g_failed;
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "main thread: %s\n", e.what());
    ++g_failed;
  }
  for (size_t t = 0; t < th.size(); ++t) th[t].join();
  std::printf("%d failures\n", g_failed.load());
  return g_failed.load() ? 1 : 0;
}
