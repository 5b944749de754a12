"""General use functions (reference `utils/utils.py:10-54`)."""
import argparse
import time


def sync(i, start_time, timestep):
    """Sleep so that a stepped simulation keeps pace with the wall clock (reference `:10-29`)."""
    if timestep > .04 or i % (int(1 / (24 * timestep))) == 0:
        elapsed = time.time() - start_time
        if elapsed < (i * timestep):
            time.sleep(timestep * i - elapsed)


def str2bool(val):
    """Converts a string into a boolean (argparse helper, reference `:33-54`)."""
    if isinstance(val, bool):
        return val
    elif val.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    elif val.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    else:
        raise argparse.ArgumentTypeError("[ERROR] in str2bool(), a Boolean value is expected")
