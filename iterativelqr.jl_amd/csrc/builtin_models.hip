// Built-in model zoo: instantiates the solve kernels for each generated model
// header (csrc/models/, produced by tools/gen_builtin_models.py) and registers
// them with the library.
#include "ilqr_device.hpp"

#include "models/model_particle.h"
#include "models/model_pendulum_euler.h"
#include "models/model_acrobot.h"
#include "models/model_car.h"
#include "models/model_car_goal.h"
#include "models/model_car_obs.h"
#include "models/model_synth32.h"
#include "models/model_synth12.h"

#ifdef ILQR_BUILTIN_ONLY      // development builds (kernel experiments): one model, a quarter of a minute instead of a whole one
#define ILQR_DEFINE_MODEL_X(M) ILQR_DEFINE_MODEL(M)         // -DILQR_BUILTIN_ONLY=Model_car
ILQR_DEFINE_MODEL_X(ILQR_BUILTIN_ONLY)
#else
ILQR_DEFINE_MODEL(Model_particle)
ILQR_DEFINE_MODEL(Model_pendulum_euler)
ILQR_DEFINE_MODEL(Model_acrobot)
ILQR_DEFINE_MODEL(Model_car)
ILQR_DEFINE_MODEL(Model_car_goal)
ILQR_DEFINE_MODEL(Model_car_obs)
ILQR_DEFINE_MODEL(Model_synth32)
ILQR_DEFINE_MODEL(Model_synth12)
#endif
