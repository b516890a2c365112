// The registration interface of the reference's GPU program, re-hosted on libmislam.so (include/mi_slam.h).
//
// Same names, argument order/meaning and error behaviour as the reference, so the caller side (Common::Main,
// TestRunner::RunSingle) does not change:
//   GetCudaIcpTransformationMatrix   source/cuda-slam/icpcuda.cuh:5-11
//   GetCudaCpdTransformationMatrix   source/cuda-slam/cpdcuda.cuh:5-17
//   GetGpuSlamResult                 source/cuda-slam/gpumain.cpp:12-38  (a Common::SlamFunc, testrunner.h:7-8)
// Results come back by value as (rotation, translation) with the out-params *iterations and *error; a device failure
// prints the message and exits with EXIT_FAILURE like checkCudaErrors (include/helper_cuda.h:567-573).
//
// The driver rules default to the reference's GPU ones (mi_icp_params_cuda_slam); `SlamRules::CpuSlam` selects the
// cpu-slam-compatible rules the parity tests use (additive translation, distance filter, no abort).
#pragma once
#include <functional>
#include <utility>

#include "configuration.h"
#include "slam_types.h"

namespace Common {
using SlamFunc = std::function<std::pair<Mat3, Vec3>(const CpuCloud&, const CpuCloud&, Configuration, int*, float*)>;
}

enum class SlamRules { CudaSlam, CpuSlam };
void SetSlamRules(SlamRules rules, float max_distance_squared = 1000.f);
void SetSlamDevice(int device);
struct mi_ctx;
mi_ctx* GetSlamContext();      // the process-wide device context the registration calls use (created on first use)

std::pair<Common::Mat3, Common::Vec3> GetCudaIcpTransformationMatrix(const std::vector<Common::Point_f>& cloudBefore,
                                                                    const std::vector<Common::Point_f>& cloudAfter, float eps,
                                                                    int maxIterations, int* iterations, float* error);

std::pair<Common::Mat3, Common::Vec3> GetCudaCpdTransformationMatrix(const std::vector<Common::Point_f>& cloudBefore,
                                                                    const std::vector<Common::Point_f>& cloudAfter, float eps,
                                                                    float weight, bool const_scale, int maxIterations, float tolerance,
                                                                    Common::ApproximationType fgt, int* iterations, float* error,
                                                                    const float& ratioOfFarField, const float& orderOfTruncation);

// nicpcuda.cuh:5-15.  batchSize only shapes the reference's thread batches; the sequential policy's semantics apply here.
std::pair<Common::Mat3, Common::Vec3> GetCudaNicpTransformationMatrix(const std::vector<Common::Point_f>& before,
                                                                     const std::vector<Common::Point_f>& after, float eps,
                                                                     int maxRepetitions, int batchSize,
                                                                     Common::ApproximationType approximationType, const int subcloudSize,
                                                                     int* repetitions, float* error);

std::pair<Common::Mat3, Common::Vec3> GetGpuSlamResult(const Common::CpuCloud& before, const Common::CpuCloud& after,
                                                      Common::Configuration configuration, int* iterations, float* error);
