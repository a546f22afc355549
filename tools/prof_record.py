"""Where a scripted recording spends its time (cProfile): python tools/prof_record.py   (32 episodes with zed_cam + cam_left_wrist, 256 without images)"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from av_aloha_amd import harness


def run():
    n = [0]
    def sink(e): n[0] += 1
    t = time.time()
    harness.record_scripted("sim_slot_insertion", 32, cameras=["zed_cam", "cam_left_wrist"], seed=0, sink=sink)
    print("32 episodes with zed_cam + cam_left_wrist:", round(time.time() - t, 1), "s;", n[0], "kept")
    t = time.time()
    harness.record_scripted("sim_slot_insertion", 256, cameras=[], seed=0, sink=sink)
    print("256 episodes without images:", round(time.time() - t, 1), "s")


cProfile.run("run()", "/tmp/rec.prof")
pstats.Stats("/tmp/rec.prof").sort_stats("cumulative").print_stats(22)
