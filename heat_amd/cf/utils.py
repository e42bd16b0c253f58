"""Mirror of the reference's cf/utils.py:5-9."""
import yaml


def load_config(config_path):
    with open(config_path, "r") as f:
        return yaml.safe_load(f)
