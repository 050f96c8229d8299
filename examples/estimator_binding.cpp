// The binding of INTEGRATION.md as code that compiles: a stand-alone `MiniEstimator` carries the members
// Estimator::optimization() reads and writes (same names and array shapes as vins_estimator/src/estimator/estimator.h:230-238,
// 283, 319, 336-347 and feature_manager.h — plain arrays and std::list instead of Eigen / ROS types) and implements
// optimization() on top of libgfbe.so exactly as a maintainer would inside the real class: vector2double() has filled para_*,
// the feature list is flattened in list order, the host owns the marginalisation prior between calls, the solved blocks and
// depths are written back. main() drives two consecutive frames (MARGIN_OLD, slide, MARGIN_OLD with the carried prior).
//   g++ -std=c++17 -I include examples/estimator_binding.cpp -L ground-fusion2_amd/csrc -lgfbe -Wl,-rpath,$PWD/ground-fusion2_amd/csrc
#include <cmath>
#include <cstdio>
#include <cstring>
#include <list>
#include <vector>

#include "gfbe.h"

constexpr int WINDOW_SIZE = GFBE_WINDOW_SIZE;
enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };

struct FeaturePerFrame { double point[3], uv[2], velocity[2], cur_td; };                       // feature_manager.h:33-70
struct FeaturePerId {                                                                           // feature_manager.h:72-99
  int feature_id, start_frame;
  std::vector<FeaturePerFrame> feature_per_frame;
  double estimated_depth;
  int estimate_flag, solve_flag;
};

struct MiniEstimator {
  // parameter blocks (estimator.h:336-342, 230-238)
  double para_Pose[WINDOW_SIZE + 1][7], para_SpeedBias[WINDOW_SIZE + 1][9], para_Ex_Pose[2][7], para_Td[1][1];
  double para_Ex_Pose_wheel[1][7], para_Ix_sx_wheel[1][1], para_Ix_sy_wheel[1][1], para_Ix_sw_wheel[1][1], para_Td_wheel[1][1];
  std::list<FeaturePerId> feature;                  // f_manager.feature (estimator.h:319)
  std::vector<gfbe_imu_preint> pre_integrations;    // packed copies of pre_integrations[1..frame_count] (estimator.h:283)
  int frame_count = WINDOW_SIZE;
  MarginalizationFlag marginalization_flag = MARGIN_OLD;
  // what replaces last_marginalization_info + last_marginalization_parameter_blocks (estimator.h:346-347)
  gfbe_ctx *gfbe_ = nullptr;
  gfbe_prior prior_{};
  std::vector<double> prior_J0_, prior_r0_;

  bool setParameter() {
    gfbe_options opt;
    gfbe_default_options(&opt);
    prior_J0_.assign((size_t)GFBE_DENSE_DIM * GFBE_DENSE_DIM, 0.0);
    prior_r0_.assign(GFBE_DENSE_DIM, 0.0);
    prior_.J0 = prior_J0_.data(); prior_.r0 = prior_r0_.data(); prior_.valid = 0;
    return gfbe_create(&gfbe_, 0, &opt) == GFBE_OK;
  }

  // Estimator::optimization() (estimator.cpp:2951-3698) on the library. Returns the library status.
  gfbe_status optimization(gfbe_summary *sum) {
    gfbe_window w;
    std::memset(&w, 0, sizeof w);
    w.frame_count = frame_count;
    std::memcpy(w.state.para_Pose, para_Pose, sizeof para_Pose);
    std::memcpy(w.state.para_SpeedBias, para_SpeedBias, sizeof para_SpeedBias);
    std::memcpy(w.state.para_Ex_Pose, para_Ex_Pose[0], sizeof para_Ex_Pose[0]);
    std::memcpy(w.state.para_Ex_Pose_wheel, para_Ex_Pose_wheel[0], sizeof para_Ex_Pose_wheel[0]);
    w.state.para_Ix_wheel[0] = para_Ix_sx_wheel[0][0]; w.state.para_Ix_wheel[1] = para_Ix_sy_wheel[0][0]; w.state.para_Ix_wheel[2] = para_Ix_sw_wheel[0][0];
    w.state.para_Td = para_Td[0][0]; w.state.para_Td_wheel = para_Td_wheel[0][0];
    // SetParameterBlockConstant decisions (estimator.cpp:3022-3161): the shipped yamls estimate none of these here
    w.ex_cam_const = w.ex_wheel_const = w.ix_wheel_const = w.td_const = w.td_wheel_const = 1;

    // f_manager.feature flattened in list order (the order that defines para_Feature[k], feature_manager.cpp:286-302)
    std::vector<int32_t> start, nobs, off, eflag;
    std::vector<double> obs, obs_td, depth;
    for (const FeaturePerId &f : feature) {
      start.push_back(f.start_frame); nobs.push_back((int32_t)f.feature_per_frame.size()); off.push_back((int32_t)obs_td.size());
      depth.push_back(f.estimated_depth); eflag.push_back(f.estimate_flag);
      for (const FeaturePerFrame &o : f.feature_per_frame) {
        const double row[7] = {o.point[0], o.point[1], o.point[2], o.uv[0], o.uv[1], o.velocity[0], o.velocity[1]};
        obs.insert(obs.end(), row, row + 7);
        obs_td.push_back(o.cur_td);
      }
    }
    const gfbe_feature_list fl = {(int32_t)start.size(), start.data(), nobs.data(), off.data(), obs.data(), obs_td.data(), depth.data(), eflag.data()};
    const int L = gfbe_feature_count(&fl), K = gfbe_visual_factor_count(&fl, 0);
    std::vector<int32_t> idx(K), ii(K), jj(K);
    std::vector<double> pi(3 * (size_t)K), pj(3 * (size_t)K), vi(2 * (size_t)K), vj(2 * (size_t)K), tdi(K), tdj(K), lam(L);
    std::vector<uint8_t> lconst(L);
    gfbe_build_visual_factors(&fl, 0, idx.data(), ii.data(), jj.data(), pi.data(), pj.data(), vi.data(), vj.data(), tdi.data(), tdj.data(), lam.data(), lconst.data());
    w.n_feature = L; w.para_Feature = lam.data(); w.feature_const = lconst.data();
    w.vis = {K, idx.data(), ii.data(), jj.data(), pi.data(), pj.data(), vi.data(), vj.data(), tdi.data(), tdj.data()};

    std::vector<int32_t> imu_frame;
    for (int i = 0; i < (int)pre_integrations.size(); i++) imu_frame.push_back(i);
    w.n_imu = (int32_t)pre_integrations.size(); w.imu = pre_integrations.data(); w.imu_frame = imu_frame.data();
    if (w.n_imu == 0) w.pose_const[0] = 1;      // (this demo has no inertial factors: hold frame 0 instead)
    w.prior = prior_.valid ? &prior_ : nullptr;

    gfbe_state out;
    const int flag = frame_count < WINDOW_SIZE ? GFBE_MARGIN_NONE : (marginalization_flag == MARGIN_OLD ? GFBE_MARGIN_OLD : GFBE_MARGIN_SECOND_NEW);
    const gfbe_status rc = gfbe_solve_window(gfbe_, &w, flag, &out, lam.data(), &prior_, sum);
    // include/gfbe.h "FAILURE CONTRACT": rc >= GFBE_BAD_INPUT — the call failed and touched no output: keep the previous state;
    // GFBE_NUMERICAL_FAILURE — the solve ran and failed, every output was written (the last accepted state and, in place, the prior
    // marginalised there): go on with them, as the reference does (it never looks at Ceres' termination_type, estimator.cpp:3377-3379)
    if (rc >= GFBE_BAD_INPUT) { std::printf("gfbe: %s\n", gfbe_last_error(gfbe_)); return rc; }
    if (rc == GFBE_NUMERICAL_FAILURE) std::printf("gfbe: linear solve failed in this window (%s)\n", gfbe_last_error(gfbe_));

    // double2vector(): `out` holds the re-anchored blocks (estimator.cpp:2515-2555)
    std::memcpy(para_Pose, out.para_Pose, sizeof para_Pose);
    std::memcpy(para_SpeedBias, out.para_SpeedBias, sizeof para_SpeedBias);
    std::memcpy(para_Ex_Pose[0], out.para_Ex_Pose, sizeof para_Ex_Pose[0]);
    std::memcpy(para_Ex_Pose_wheel[0], out.para_Ex_Pose_wheel, sizeof para_Ex_Pose_wheel[0]);
    // FeatureManager::setDepth (feature_manager.cpp:249-267)
    std::vector<int32_t> solve_flag(fl.n, 0);
    gfbe_set_depth(&fl, lam.data(), depth.data(), solve_flag.data());
    int q = 0;
    for (FeaturePerId &f : feature) { f.estimated_depth = depth[q]; f.solve_flag = solve_flag[q]; q++; }
    return rc;
  }

  // the part of slideWindow() (estimator.cpp:3700-3760) this demo needs after MARGIN_OLD: blocks and tracks move one frame back
  void slideWindowOld(double new_x) {
    for (int i = 0; i < WINDOW_SIZE; i++) { std::memcpy(para_Pose[i], para_Pose[i + 1], sizeof para_Pose[i]); std::memcpy(para_SpeedBias[i], para_SpeedBias[i + 1], sizeof para_SpeedBias[i]); }
    para_Pose[WINDOW_SIZE][0] = new_x;
    for (auto it = feature.begin(); it != feature.end();) {      // removeBack: drop the observation of the departed frame
      if (it->start_frame != 0) { it->start_frame--; ++it; continue; }
      it->feature_per_frame.erase(it->feature_per_frame.begin());
      if (it->feature_per_frame.empty()) it = feature.erase(it); else ++it;
    }
  }
};

int main() {
  MiniEstimator e;
  std::memset(e.para_Pose, 0, sizeof e.para_Pose); std::memset(e.para_SpeedBias, 0, sizeof e.para_SpeedBias);
  std::memset(e.para_Ex_Pose, 0, sizeof e.para_Ex_Pose); std::memset(e.para_Ex_Pose_wheel, 0, sizeof e.para_Ex_Pose_wheel);
  e.para_Ex_Pose[0][6] = e.para_Ex_Pose_wheel[0][6] = 1.0;
  e.para_Ix_sx_wheel[0][0] = e.para_Ix_sy_wheel[0][0] = e.para_Ix_sw_wheel[0][0] = 1.0;
  e.para_Td[0][0] = e.para_Td_wheel[0][0] = 0.0;
  if (!e.setParameter()) { std::printf("no usable GPU (GFBE_NO_DEVICE): there is no CPU fallback\n"); return 0; }
  // a camera moving along x (0.1 m per keyframe, identity extrinsic), 40 landmarks each seen in up to 11 consecutive frames
  auto truth_x = [](int k) { return 0.1 * k; };
  for (int i = 0; i <= WINDOW_SIZE; i++) { e.para_Pose[i][0] = truth_x(i) + 0.004 * ((i * 7) % 5 - 2); e.para_Pose[i][6] = 1.0; }
  for (int l = 0; l < 40; l++) {
    FeaturePerId f{l, l % 4, {}, 0.0, 0, 0};
    const double X = -1.5 + 0.12 * l, Y = 0.25 * (l % 3) - 0.25, Z = 4.0 + 0.4 * (l % 5);
    f.estimated_depth = Z * 1.08;
    for (int fr = f.start_frame; fr <= WINDOW_SIZE && fr < f.start_frame + 9 + l % 3; fr++)
      f.feature_per_frame.push_back({{(X - truth_x(fr)) / Z, Y / Z, 1.0}, {0, 0}, {0, 0}, 0.0});
    e.feature.push_back(f);
  }
  for (int call = 0; call < 2; call++) {
    gfbe_summary sum;
    const gfbe_status rc = e.optimization(&sum);
    if (rc > GFBE_NO_CONVERGENCE) return 1;
    std::printf("optimization() #%d: status %d, %d iterations, cost %.3e -> %.3e, prior %s (n = %d, %d blocks)\n", call + 1, (int)rc, (int)sum.iterations,
                sum.initial_cost, sum.final_cost, e.prior_.valid ? "valid" : "none", (int)e.prior_.n, (int)e.prior_.n_blocks);
    if (sum.final_cost > sum.initial_cost) return 1;
    e.slideWindowOld(truth_x(WINDOW_SIZE + 1) + 0.003);
  }
  gfbe_destroy(e.gfbe_);
  return 0;
}
