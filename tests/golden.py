"""Golden vectors produced by the reference itself (oracle/_ref/ref_harness, the pinned -ffp-contract=off build of the
unmodified RawHash2 sources).  Inputs are regenerated from seeds; only the expected PAF text is committed."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def cases():
    with open(os.path.join(GOLD, "cases.json")) as f:
        return json.load(f)


def ava_cases():
    with open(os.path.join(GOLD, "ava_cases.json")) as f:
        return json.load(f)


def repeat_cases():
    with open(os.path.join(GOLD, "repeat_cases.json")) as f:
        return json.load(f)


def build_repeat_case(case, directory, lib):
    from repeat_workload import RepeatWorkload
    return RepeatWorkload(directory, lib, **case["workload"])


def build_ava_case(case, directory, lib):
    from conftest import AvaWorkload
    return AvaWorkload(directory, lib, **case["workload"])


def build_case(case, directory, lib):
    from conftest import Workload
    return Workload(directory, lib, **case["workload"], build_index=not case.get("gpu_only"), no_adaptive=bool(case.get("no_adaptive")))


def expected_paf(case):
    with open(os.path.join(GOLD, case["name"] + ".paf")) as f:
        return [l.rstrip("\n") for l in f]
