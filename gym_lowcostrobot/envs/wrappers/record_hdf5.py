"""Import path of the reference (`from gym_lowcostrobot.envs.wrappers.record_hdf5 import RecordHDF5Wrapper`, examples/hdf5_record.py:6);
the wrapper is gym_lowcostrobot_amd.recorder.RecordHDF5Wrapper (same constructor, file naming and dataset layout:
envs/wrappers/record_hdf5.py:52-61,111)."""
from gym_lowcostrobot_amd.recorder import RecordHDF5Wrapper  # noqa: F401

__all__ = ["RecordHDF5Wrapper"]
