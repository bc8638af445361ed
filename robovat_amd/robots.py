"""RobotCommand (``robovat/robots/robot_command.py:8-34``)."""


class RobotCommand(object):
    def __init__(self, component, command_type, arguments=None):
        self.component = component
        self.command_type = command_type
        self.arguments = arguments or {}
