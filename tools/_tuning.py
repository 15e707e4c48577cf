"""The A/B kernel policies (epn_set_kernel_policy 0x100 | cfg .. 0x400 | cfg) exist only in libraries built with
-DEPN_TUNING; the tools that use them call use_tuning_lib() BEFORE importing epn_pointcloud_amd._lib's library."""
import os
import sys


def use_tuning_lib():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    if os.environ.get("EPN_LIB"):          # an A/B build named by the caller (EPN_BUILD_TAG / EPN_EXTRA_FLAGS) wins
        return os.environ["EPN_LIB"]
    lib = os.path.join(root, "epn_pointcloud_amd", "libepn_so3conv_tuning.so")
    if not os.path.exists(lib):
        from epn_pointcloud_amd import build
        lib = build.build(tuning=True)
    os.environ["EPN_LIB"] = lib
    return lib
