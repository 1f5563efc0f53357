from .kilobot import Kilobot, MotorKilobot, PhototaxisKilobot, SimplePhototaxisKilobot, SimpleVelocityControlKilobot,\
    SimpleAccelerationControlKilobot
from .body import Body, Circle, World
from .light import Light, SinglePositionLight, CircularGradientLight, GradientLight, MomentumLight, CompositeLight
