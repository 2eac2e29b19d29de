"""Cost of interpreted (user-defined) kinds against the built-in ones on the same 1920x1080 room."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import custom_kinds as ck  # noqa: E402,F401
from test_custom_kinds import room, EXTRA  # noqa: E402
from madarch_amd import _binding as B  # noqa: E402
hb = B.hip_binding()
for custom, extra, jit, name in ((False, (), 0, "built-in kinds"), (True, (), 0, "the same kinds, interpreted"), (True, (), 1, "the same kinds, hiprtc"),
                                 (False, EXTRA, 0, "built-in + torus, ripple, 2 capsules, interpreted"), (False, EXTRA, 1, "built-in + torus, ripple, 2 capsules, hiprtc")):
    R = room(hb, custom, W=1920, H=1080, extra=extra)
    R.Set_Option(B.OPT_GBUFFER, 0)
    R.Set_Option(B.OPT_JIT, jit)
    t = time.perf_counter(); R.Render(); R.Finish(); first = time.perf_counter() - t
    for _ in range(3): R.Render()
    R.Finish(); t = time.perf_counter()
    for _ in range(5): R.Render()
    R.Finish(); dt = (time.perf_counter() - t) / 5
    print("%-52s %8.2f ms/frame  %6.0f Mpix/s   (first frame %.2f s)" % (name, dt * 1e3, 1920 * 1080 / dt / 1e6, first), flush=True)
