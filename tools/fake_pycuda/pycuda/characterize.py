def sizeof(type_name, preamble=""):
    return 48
