// main.cc -- command line with the reference's ten flags and defaults (CSPM/main.cc:23-34) and its flow
// (main.cc:57-139): read the pair, construct the plane cost (timed), run PatchMatch, print "Total Time", write the
// two 8-bit maps.  Runs on the GPU through the host layer.  Extra flags: --seed --schedule --device --iters.
#include "commfunc.h"
#include "cs_patchmatch.h"
#include "get_method.h"
#include "plane_cost/cspc.h"
#include "plane_cost/grd_pc.h"
#include "plane_cost/pre_cs_pc.h"
#include "plane_cost/pre_ss_pc.h"
#include "pfm_io.h"

#include <atomic>
#include <fstream>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

// images
DEFINE_string(l_img_file, "l_img.png", "left input image (8-bit PNG / PPM / PGM)");
DEFINE_string(r_img_file, "r_img.png", "right input image");
DEFINE_string(l_dis_file, "l_dis.png", "left disparity map to write (8-bit)");
DEFINE_string(r_dis_file, "r_dis.png", "right disparity map to write (8-bit)");
// matching
DEFINE_int32(max_dis, 0, "disparity search range");
DEFINE_int32(dis_scale, 0, "factor applied to disparities before 8-bit quantisation");
DEFINE_string(cc_name, "CCName", "matching cost: GRD | CEN");
DEFINE_string(pc_name, "PRE", "plane cost family: PRE = PreSSPC / PreCSPC over --cc_name's cost volumes (the reference's main.cc); "
                              "IMG = GrdPC / CSPC, the volume-free colour + gradient costs (main.cc:106-107, commented out there)");
DEFINE_bool(use_cs, false, "cross-scale aggregation over a 5-level pyramid (PreCSPC) instead of PreSSPC");
DEFINE_bool(use_pp, false, "left-right check, hole filling and weighted median afterwards");
DEFINE_double(reg_lambda, 0.0, "cross-scale regularisation weight");
// not in the reference
DEFINE_int32(seed, 12345, "random seed (the reference uses the wall clock)");
DEFINE_string(schedule, "raster", "spatial propagation: raster (the reference's sweep) | redblack");
DEFINE_int32(device, 0, "GPU index");
DEFINE_int32(iters, 3, "PatchMatch iterations (3 in the reference, main.cc:93)");
DEFINE_string(l_disp_pfm, "", "also write the left sub-pixel disparity map as float32 PFM: the unquantised plane disparity a*x+b*y+c, "
                              "BEFORE post-processing (--use_pp changes the 8-bit maps only, as in the reference)");
DEFINE_string(r_disp_pfm, "", "also write the right sub-pixel disparity map as float32 PFM (see --l_disp_pfm)");
DEFINE_string(batch_list, "", "text file, one stereo pair per line: l_img r_img l_dis r_dis [l_pfm r_pfm]; all pairs run with the "
                              "matching flags of this command line on one device context (buffers are reused between pairs). A pair "
                              "that fails is reported and the batch goes on; the exit code is non-zero if any pair failed");
DEFINE_int32(in_flight, 2, "with --batch_list: stereo pairs in flight per GPU (2 measured best on MI355X, 3 and 4 within 1.5 %).  Each is a worker thread with its own device context (one HIP stream): "
                           "it decodes its pair's PNGs, runs it and encodes the maps while the other workers' kernels keep the GPU busy (the raster "
                           "sweep of one pair leaves most CUs idle).  With 2 or more the sweep runs four-wavefront workgroups (CSPM_OPT_SWEEP_FOLD)");
DEFINE_string(devices, "", "with --batch_list: GPUs to spread the pairs over: a comma-separated list of indices (an index may repeat: that many "
                           "worker sets on that GPU) or `all`; empty = --device.  Pairs are independent: no data moves between GPUs");
DEFINE_bool(batch_skip_existing, false, "with --batch_list: skip the pairs whose output maps already exist (restart an interrupted batch)");
DEFINE_bool(quiet, false, "print errors and the batch summary only");

namespace {
const int kWindow = 35;  // main.cc:94
const int kScales = 5;   // main.cc:100

struct PairFiles {
  string l_img, r_img, l_dis, r_dis, l_pfm, r_pfm;
};

// One stereo pair on its way through the flow of main.cc:57-139, cut into the four stages a batch worker overlaps: load (decode the
// files), begin (construct the plane cost, enqueue PatchMatch on the calling thread's device slot), finish (wait, fetch the maps,
// release the cost object -- in batch mode its context is parked for the next pair), write (encode the maps).  `log` collects what the
// reference prints (a batch worker's lines are written out in one piece when its pair is done).
struct PairRun {
  PairFiles files;
  int line_no;
  Mat left, right;
  std::unique_ptr<IPlaneCost> cost;
  std::unique_ptr<CSPatchMatch> matcher;
  std::vector<double> pfm[kViewNum];
  double t0;
  int rc;
  std::ostringstream log;
  PairRun(const PairFiles &f, int line) : files(f), line_no(line), t0(0.0), rc(EXIT_SUCCESS) {}
};

void load(PairRun &p) {
  p.left = imread(p.files.l_img, CV_LOAD_IMAGE_COLOR);
  p.right = imread(p.files.r_img, CV_LOAD_IMAGE_COLOR);
  if (p.left.empty() || p.right.empty()) {
    // the reference waits for a key press here (main.cc:70-75); a batch tool must not
    p.log << "Error: can not open image\n";
    p.rc = EXIT_FAILURE;
  }
}

void begin(PairRun &p, CCMethod *cost_fn) {
  if (p.rc != EXIT_SUCCESS) return;
  try {
    p.t0 = static_cast<double>(getTickCount());
    IPlaneCost *pc;
    if (FLAGS_pc_name == "IMG")
      pc = FLAGS_use_cs ? static_cast<IPlaneCost *>(new CSPC(p.left, p.right, FLAGS_max_dis, kWindow, kScales, FLAGS_reg_lambda))
                        : static_cast<IPlaneCost *>(new GrdPC(p.left, p.right, FLAGS_max_dis, kWindow));
    else
      pc = FLAGS_use_cs ? static_cast<IPlaneCost *>(new PreCSPC(p.left, p.right, FLAGS_max_dis, kWindow, kScales, cost_fn, FLAGS_reg_lambda))
                        : static_cast<IPlaneCost *>(new PreSSPC(p.left, p.right, FLAGS_max_dis, kWindow, cost_fn));
    p.cost.reset(pc);  // released on every path, exceptions included (batch mode goes on)
    p.matcher.reset(new CSPatchMatch(p.left, p.right, FLAGS_max_dis, FLAGS_dis_scale));
    p.matcher->set_seed(static_cast<uint64_t>(FLAGS_seed));
    p.matcher->set_schedule(FLAGS_schedule == "redblack" ? 1 : 0);
    p.matcher->PatchMatchBegin(FLAGS_iters, p.cost.get(), FLAGS_use_pp);
  } catch (const std::exception &e) {  // a bad pair must not take the batch down
    p.log << "Error: " << e.what() << "\n";
    p.rc = EXIT_FAILURE;
    p.cost.reset();
  }
}

void finish(PairRun &p) {
  if (p.rc == EXIT_SUCCESS) {
    try {
      p.matcher->PatchMatchEnd();
      const double seconds = (static_cast<double>(getTickCount()) - p.t0) / getTickFrequency();
      if (!FLAGS_quiet)
        p.log << "--------------------------------------------------------\n"
              << "Total Time: " << seconds << "\n"
              << "--------------------------------------------------------\n";
      const string *pfm[kViewNum] = {&p.files.l_pfm, &p.files.r_pfm};
      for (int v = 0; v < kViewNum; ++v)
        if (!pfm[v]->empty()) p.matcher->disparity(v == 0 ? kLeft : kRight, &p.pfm[v]);  // reads the cost object's context: before it goes
    } catch (const std::exception &e) {
      p.log << "Error: " << e.what() << "\n";
      p.rc = EXIT_FAILURE;
    }
  }
  p.cost.reset();
}

void write(PairRun &p) {
  if (p.rc != EXIT_SUCCESS) return;
  bool written = imwrite(p.files.l_dis, p.matcher->dis(kLeft)) && imwrite(p.files.r_dis, p.matcher->dis(kRight));
  const string *pfm[kViewNum] = {&p.files.l_pfm, &p.files.r_pfm};
  for (int v = 0; v < kViewNum && written; ++v)
    if (!pfm[v]->empty()) written = WritePFM(*pfm[v], p.pfm[v].data(), p.left.cols, p.left.rows);
  if (!written) {
    p.log << "Error: can not write disparity maps\n";
    p.rc = EXIT_FAILURE;
  }
}

struct BatchJob {
  PairFiles files;
  int line_no;
};

// --devices: "" -> {--device}; "all" -> every GPU; "0,1,1" -> those indices (repeats allowed).  Empty result = bad flag.
std::vector<int> parse_devices() {
  std::vector<int> out;
  const int n = DeviceSlot::device_count();
  if (FLAGS_devices.empty()) {
    if (FLAGS_device >= 0 && FLAGS_device < n) out.push_back(FLAGS_device);
    return out;
  }
  if (FLAGS_devices == "all") {
    for (int d = 0; d < n; ++d) out.push_back(d);
    return out;
  }
  std::istringstream is(FLAGS_devices);
  string tok;
  while (std::getline(is, tok, ',')) {
    char *end = NULL;
    const long d = std::strtol(tok.c_str(), &end, 10);
    if (tok.empty() || *end || d < 0 || d >= n) return std::vector<int>();
    out.push_back(static_cast<int>(d));
  }
  return out;
}

// The batch: pairs are independent units.  One worker thread per (entry of --devices, slot of --in_flight), each with
// its own DeviceSlot -- GPU index, parked context whose buffers the next pair of the worker takes over, whether it shares its GPU -- and
// its own CCMethod object; the workers pull pairs from one queue.  A worker blocks on its own pair only (cspm_ctx = one HIP stream), so
// file decoding / encoding and the serial phases of one pair overlap with the kernels of the others.  Results do not depend on the
// worker or GPU a pair lands on (same seed, same kernels): the maps equal the one-pair-at-a-time run bit for bit.
int run_batch(const std::vector<BatchJob> &jobs, int skipped, int bad_lines) {
  const std::vector<int> devices = parse_devices();
  if (devices.empty()) {
    cout << "Error: --devices must be `all` or a comma-separated list of GPU indices below " << DeviceSlot::device_count()
         << " (--device likewise); this node has " << DeviceSlot::device_count() << " usable GPU(s)\n";
    return EXIT_FAILURE;
  }
  const int per_gpu = std::max(1, FLAGS_in_flight);
  std::vector<int> on_gpu(DeviceSlot::device_count() > 0 ? DeviceSlot::device_count() : 1, 0);  // pairs in flight per physical GPU
  for (size_t i = 0; i < devices.size(); ++i) on_gpu[devices[i]] += per_gpu;
  std::atomic<size_t> next(0);
  std::atomic<int> failed(bad_lines), done(0);
  std::atomic<long long> sweep_fallbacks(0), volume_fallbacks(0);
  std::mutex out_mutex;
  const double t0 = static_cast<double>(getTickCount());
  auto worker = [&](int device) {
    DeviceSlot slot(device, /*keep_context=*/true, /*shared_gpu=*/on_gpu[device] >= 2);
    DeviceSlot::Use use(slot);
    const std::unique_ptr<CCMethod> cost_fn(GetCCType(FLAGS_cc_name));  // NULL for unknown names, rejected by the cost constructors
    // the next pair of the queue, decoded; a pair whose files cannot be read is reported at once and the worker moves on
    auto report = [&](PairRun &p) {
      if (p.rc != EXIT_SUCCESS) {
        ++failed;
        p.log << "Pair FAILED (line " << p.line_no << "): " << p.files.l_img << " " << p.files.r_img << "\n";
      }
      ++done;
      std::lock_guard<std::mutex> lock(out_mutex);
      cout << p.log.str() << std::flush;
    };
    auto next_loaded = [&]() -> std::unique_ptr<PairRun> {
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= jobs.size()) return std::unique_ptr<PairRun>();
        std::unique_ptr<PairRun> p(new PairRun(jobs[k].files, jobs[k].line_no));
        if (!FLAGS_quiet) p->log << "Load Image: " << p->files.l_img << " " << p->files.r_img << "\n";
        load(*p);
        if (p->rc == EXIT_SUCCESS) return p;
        report(*p);
      }
    };
    // software pipeline: while pair k is on the GPU the worker decodes pair k+1; as soon as k's maps are on the host, k+1 is
    // enqueued (on the context k just parked) and only then are k's maps encoded and written -- the worker's stream idles for the
    // download and the upload only, not for the files
    try {
      std::unique_ptr<PairRun> cur = next_loaded();
      if (cur) begin(*cur, cost_fn.get());
      while (cur) {
        std::unique_ptr<PairRun> nxt = next_loaded();
        finish(*cur);
        if (nxt) begin(*nxt, cost_fn.get());
        write(*cur);
        report(*cur);
        cur = std::move(nxt);
      }
    } catch (const std::exception &e) {  // the stages catch per pair; what still gets here (out of memory on the host ...) ends this worker, not the process
      ++failed;
      std::lock_guard<std::mutex> lock(out_mutex);
      cout << "Error: batch worker on GPU " << device << " stopped: " << e.what() << "\n" << std::flush;
    }
    slot.release();
    sweep_fallbacks += slot.sweep_fallbacks();
    volume_fallbacks += slot.volume_fallbacks();
  };
  std::vector<std::thread> threads;
  const size_t want = devices.size() * static_cast<size_t>(per_gpu);
  for (size_t t = 0; t < std::min(want, std::max<size_t>(jobs.size(), 1)); ++t) threads.emplace_back(worker, devices[t % devices.size()]);
  for (size_t t = 0; t < threads.size(); ++t) threads[t].join();
  const double seconds = (static_cast<double>(getTickCount()) - t0) / getTickFrequency();
  cout << "Batch: " << done.load() + bad_lines << " pairs in " << seconds << " s, " << failed.load() << " failed, " << skipped << " skipped\n";
  if (!FLAGS_quiet) {
    cout << "Batch workers: " << threads.size() << " (" << devices.size() << " GPU entr" << (devices.size() == 1 ? "y" : "ies") << " x " << per_gpu
         << " in flight), " << (done.load() ? seconds * 1e3 / done.load() : 0.0) << " ms per pair end to end, files included\n";
    cout << "Batch fallbacks: " << sweep_fallbacks.load() << " raster sweeps repeated after a hand-over timeout, " << volume_fallbacks.load()
         << " optional volumes given up\n";
  }
  return failed.load() ? EXIT_FAILURE : EXIT_SUCCESS;
}

int run() {
  DevicePlaneCost::device = FLAGS_device;
  if (FLAGS_use_pp && !(FLAGS_l_disp_pfm.empty() && FLAGS_r_disp_pfm.empty()) && !FLAGS_quiet)
    cout << "Note: the PFM maps hold the plane disparities before post-processing\n";
  if (FLAGS_batch_list.empty()) {
    const std::unique_ptr<CCMethod> cost_fn(GetCCType(FLAGS_cc_name));  // NULL for unknown names, rejected by the cost constructors
    if (!FLAGS_quiet) cout << "Load Image: " << FLAGS_l_img_file << " " << FLAGS_r_img_file << "\n";
    PairRun p(PairFiles{FLAGS_l_img_file, FLAGS_r_img_file, FLAGS_l_dis_file, FLAGS_r_dis_file, FLAGS_l_disp_pfm, FLAGS_r_disp_pfm}, 0);
    load(p);
    begin(p, cost_fn.get());
    finish(p);
    write(p);
    cout << p.log.str();
    return p.rc;
  }
  std::ifstream list(FLAGS_batch_list.c_str());
  if (!list) {
    cout << "Error: can not open batch list " << FLAGS_batch_list << "\n";
    return EXIT_FAILURE;
  }
  std::vector<BatchJob> jobs;
  string line;
  int bad_lines = 0, skipped = 0, line_no = 0;
  while (std::getline(list, line)) {
    ++line_no;
    std::istringstream is(line);
    PairFiles f;
    if (!(is >> f.l_img)) continue;  // blank line
    if (f.l_img[0] == '#') continue;
    if (!(is >> f.r_img >> f.l_dis >> f.r_dis)) {
      cout << "Error: batch list line " << line_no << " needs l_img r_img l_dis r_dis: " << line << "\n";
      ++bad_lines;
      continue;
    }
    is >> f.l_pfm >> f.r_pfm;
    if (FLAGS_batch_skip_existing && std::ifstream(f.l_dis.c_str()).good() && std::ifstream(f.r_dis.c_str()).good()) {
      ++skipped;
      continue;
    }
    jobs.push_back(BatchJob{f, line_no});
  }
  return run_batch(jobs, skipped, bad_lines);
}
}  // namespace

int main(int argc, char **argv) {
  // HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; two pair streams that share a queue run one after the other
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  gflags::ParseCommandLineFlags(&argc, &argv, true);
  if (!FLAGS_quiet) cout << "PatchMatch Stereo Matching (MI355X)" << endl;
  try {
    return run();
  } catch (const std::exception &e) {
    cout << "Error: " << e.what() << endl;
    return EXIT_FAILURE;
  }
}
