from .device_input import DeviceInputPipeline  # noqa: F401
