"""string_ops.py:6-15 of the reference: option strings that mean True."""


def check_boolean(input_string):
    return str(input_string).lower() in ("yes", "true", "1")
