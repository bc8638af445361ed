"""RobotCommand (``robovat/robots/robot_command.py:8-34``): what a robot facade hands to
``Simulator.receive_robot_commands`` (simulator.py:226-244) -- the name of a component of the simulator, the
name of one of its methods and the keyword arguments of the call."""


class RobotCommand(object):
    def __init__(self, component, command_type, arguments=None):
        self.component = component
        self.command_type = command_type
        self.arguments = arguments or {}
