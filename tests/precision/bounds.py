"""ABSOLUTE error bounds per compute precision for the synthetic O(1) test networks (tests/golden/cases.py NET_CASES), against the
reference's golden outputs.  They are fixed numbers, not "N x what this implementation measured":

* f32, f32x3: the north_star tolerances (heat maps 1e-3, 3D points 1e-4 m, peak indices identical).
* f16, bf16: the rounding-point model of tests/precision/emulate.py (a CPU computation that only knows WHERE the HIP path rounds to
  16 bits - 137 stored tensors, 139 folded weight sets - not how its kernels work) predicts heat max 1.8e-3 / mean 2.1e-4 for fp16 and
  8x that (three mantissa bits fewer) for bf16 on these networks; the bounds are twice the prediction, rounded.  tests/test_precision_emulation.py
  keeps the model inside them on the CPU, tests/test_gpu_net.py holds the device to the same numbers.
  No selection of layers brings a 16-bit configuration under the 1e-3 heat bar: tests/golden/precision_attribution.json shows the error
  variance spread over all rounding points (largest single share 4 %), which is why the configuration that meets the bar at MFMA speed is
  the three-term split f32x3, not a mixed-precision fp16 one.
* f32mix (fp32 skip stream with three-term products, single-term fp16 residual branches, fp16 hourglass levels <= 16 x 16): meets the
  HEAT bar only.  The rounding-point model predicts heat max 4.1e-4 / 5.0e-4 on the two golden networks (bound 7.5e-4 there; 9e-4 for
  the worst of 16 further frames: measured range 3e-4 ... 6.5e-4); depth / centre maps within 4e-3; peak sets agree to Jaccard >= 0.99
  (NOT identical); 3D points of the drop-in sequence within 5e-3 m (50x the 1e-4 m bar: a centroid moves by up to 2e-2 px and the
  depth map by 4e-3).  float32x3 is the fastest configuration inside all three north_star tolerances.  The numbers hold for the
  synthetic weight family the plan was derived on - tests/golden/precision_families.json prices others.
"""

BOUNDS = {
    "f32":   {"heat_max": 1e-3, "heat_mean": 1e-4, "depth_max": 1e-3, "centers_max": 1e-3, "jaccard_min": 1.0, "p_C_max_m": 1e-4, "p_C_mean_m": 1e-4},
    "f32x3": {"heat_max": 1e-3, "heat_mean": 1e-4, "depth_max": 1e-3, "centers_max": 1e-3, "jaccard_min": 1.0, "p_C_max_m": 1e-4, "p_C_mean_m": 1e-4},
    "f32mix": {"heat_max": 7.5e-4, "heat_max_any_frame": 9e-4, "heat_mean": 1e-4, "depth_max": 4e-3, "centers_max": 4e-3, "jaccard_min": 0.99,
               "keypoint_max_px": 2e-2, "p_C_max_m": 5e-3},
    "f16":   {"heat_max": 4e-3, "heat_mean": 5e-4, "depth_max": 2e-2, "depth_mean": 3.5e-3, "centers_max": 2e-2, "centers_mean": 3e-3,
              "jaccard_min": 0.98, "p_C_max_m": 2e-2, "p_C_mean_m": 4e-3},
    # bf16: a rounded pixel can flip on the noise-like depth map of a random-weight network (0.5 m at one peak of the fixture), so only
    # the MEAN 3D error at common peaks is bounded
    "bf16":  {"heat_max": 3.2e-2, "heat_mean": 4e-3, "depth_max": 0.16, "depth_mean": 2.8e-2, "centers_max": 0.16, "centers_mean": 2.4e-2,
              "jaccard_min": 0.90, "p_C_max_m": None, "p_C_mean_m": 3e-2},
}
