from .noise_schedule import NoiseScheduleVP  # noqa: F401
