// device_plane_cost.h -- common implementation of PreSSPC / PreCSPC / GrdPC / CSPC above the C ABI (include/cspm.h), and
// DeviceSlot: where a host thread's cost objects live (which GPU, which options, which parked context).
#pragma once
#include <vector>
#include "../cc_method.h"
#include "i_plane_cost.h"

// One host thread's share of one GPU.  The constructors of PreSSPC / PreCSPC / GrdPC / CSPC keep the reference's signatures
// (pre_cs_pc.h:21-23, pre_ss_pc.h:20-22, grd_pc.h:27-29, cspc.h:21-23), which have no room for a device argument; a thread says where
// its objects go by making a slot current (DeviceSlot::Use, RAII, thread-local), or passes the slot to the extended constructors of
// DevicePlaneCost.  Nothing in a slot is shared: two threads with two slots run two GPUs -- or two pair streams of one GPU -- through
// the reference-shaped classes at the same time (one cspm_ctx = one HIP stream, include/cspm.h).  A slot must outlive the objects made
// through it and is used by one thread at a time.  Threads that never name a slot share the process-wide DEFAULT slot, configured by the
// statics DevicePlaneCost::device / keep_context (the single-threaded reference flow, main.cc:57-139): set those before the first object.
class DeviceSlot {
 public:
  // keep_context: a destroyed cost object parks its cspm_ctx (device buffers included) here for the next object of the slot, so a stream
  // of equally sized pairs allocates once.  shared_gpu: other slots keep pairs in flight on the same GPU -- the contexts opened through
  // this one then get CSPM_OPT_SWEEP_FOLD (four-wavefront sweep workgroups that leave the other pairs' kernels room, include/cspm.h).
  explicit DeviceSlot(int device = 0, bool keep_context = false, bool shared_gpu = false)
      : device_(device), keep_(keep_context), shared_gpu_(shared_gpu), parked_(NULL), sweep_fallbacks_(0), volume_fallbacks_(0) {}
  ~DeviceSlot() { release(); }
  int device() const { return device_; }
  bool keep_context() const { return keep_; }
  bool shared_gpu() const { return shared_gpu_; }
  void release();  // destroy the parked context, if any
  // sums over the contexts this slot has seen: raster sweeps repeated after a hand-over timeout / optional volumes given up (include/cspm.h)
  long long sweep_fallbacks() const { return sweep_fallbacks_; }
  long long volume_fallbacks() const { return volume_fallbacks_; }

  // makes `slot` the calling thread's current slot for the lifetime of the Use object (nestable)
  class Use {
   public:
    explicit Use(DeviceSlot &slot);
    ~Use();
   private:
    DeviceSlot *prev_;
    Use(const Use &);
  };
  static DeviceSlot &current();  // the calling thread's slot; the default slot when no Use is active
  static int device_count();     // usable GPUs (0 when there is none: nothing in this layer runs on the CPU)

 private:
  friend class DevicePlaneCost;
  DeviceSlot(const DeviceSlot &);
  cspm_ctx *take();            // the parked context (ownership moves to the caller) or NULL
  bool park(cspm_ctx *ctx);    // false: not a keeping slot / already holds one -- the caller destroys ctx
  int device_;
  bool keep_;
  bool shared_gpu_;
  cspm_ctx *parked_;
  long long sweep_fallbacks_, volume_fallbacks_;
};

class DevicePlaneCost : public IPlaneCost, public IDevicePlaneCost {
 public:
  // scale_num == 0: PreSSPC (pre_ss_pc.cc:12-65); >= 1: PreCSPC (pre_cs_pc.cc:12-115).  cc_method is borrowed.
  // slot == NULL: the calling thread's current slot (DeviceSlot::current()).
  DevicePlaneCost(const Mat &l_img, const Mat &r_img, int max_disp, int wnd_size, int scale_num, CCMethod *cc_method,
                  double reg_lambda, DeviceSlot *slot = NULL);
  // the volume-free variants, no CCMethod: scale_num == 0: GrdPC (grd_pc.cc:11-66); >= 1: CSPC (cspc.cc:11-93)
  DevicePlaneCost(const Mat &l_img, const Mat &r_img, int max_disp, int wnd_size, int scale_num, double reg_lambda, DeviceSlot *slot = NULL);
  ~DevicePlaneCost();
  virtual double GetPlaneCost(const int &ref_x, const int &ref_y, const Plane &plane, const RefView &view) const;
  virtual cspm_ctx *device_ctx() const { return ctx_; }
  // the DEFAULT slot's settings (threads without a DeviceSlot::Use): GPU index (the CLI's --device) and batch mode
  static int device;
  static bool keep_context;
  static void release_kept_context();
  // is this context still owned by a live (or parked) DevicePlaneCost?  CSPatchMatch borrows the context of the cost object it
  // ran on (planes(), disparity()) and must not touch it once that object is gone.  Thread-safe.
  static bool is_live(const cspm_ctx *ctx);
  static void adopt(cspm_ctx *ctx);   // a context owned by someone else (CSPatchMatch's own, for a foreign IPlaneCost) joins / leaves
  static void disown(cspm_ctx *ctx);  // the registry

 private:
  DevicePlaneCost(const DevicePlaneCost &);
  void upload_foreign(CCMethod *cc, int view, int level);
  void open_context(const Mat &l_img, const Mat &r_img);
  void close_context();  // counters to the slot; park the context there or destroy it
  cspm_ctx *ctx_;
  DeviceSlot *slot_;
  long long base_sweep_fallbacks_, base_volume_fallbacks_;  // the context's counters when this object took it over
};
