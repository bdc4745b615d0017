from .tasks import DetectionModel, parse_model, yaml_model_load, guess_model_scale, guess_model_task  # noqa: F401
